// The issue schedule of HeadStream (ldm_pipes.h) as pure constexpr C++ — no HIP — so that its lgkmcnt bookkeeping can be
// replayed on the CPU (tests/cpu_sched_check.cpp): which LDS operations a step issues, in which order, and how many of
// them are younger than the fragment a step waits for.  LDS operations of one wave complete in order, so
// "s_waitcnt lgkmcnt(n)" = "everything but the n youngest operations has completed".
#pragma once

namespace ldm_sched {

template <bool LEAN>
struct HeadSched {
  static constexpr int KS = 29, NT = 6, NIT = KS * NT, PF = 6;
  static constexpr int SYNC = KS - PF;   // local step of the per-tile barrier
  static constexpr int EPI0 = 10;        // first local step of the previous tile's epilogue
  // local step at which the LEAN variant reads the bias of the PREVIOUS tile (consumed by that tile's epilogue at EPI0):
  // PF + 1 steps ahead, so that the counted wait of step EPI0 - 1 covers it
  static constexpr int BIAS_LEAN = EPI0 - PF - 1;

  static constexpr bool tile_has_bias(int j) { return j < 2 || j >= 4; }  // k0 k1 | v0 v1 | q0 q1
  // bias reads (4 x ds_read_b128) issued at global step s, BEFORE the step's fragment read
  static constexpr int bias_at(int s) {
    const int j = s / KS, it = s % KS;
    if (LEAN) return ((it == BIAS_LEAN && j >= 1 && tile_has_bias(j - 1)) || (j == NT - 1 && it == SYNC)) ? 4 : 0;
    return (it == SYNC && j + 1 < NT && tile_has_bias(j + 1)) ? 4 : 0;
  }
  // ds_write_b128 issued at step s, AFTER the step's fragment read (epilogue of the previous tile when that was a K or V
  // tile: two stores, slices 2 and 5)
  static constexpr int writes_at(int s) {
    const int j = s / KS, it = s % KS;
    return (j >= 1 && j <= 4 && (it == EPI0 + 2 || it == EPI0 + 5)) ? 1 : 0;
  }
  // bias reads issued by the head prologue in front of the first PF fragment reads
  static constexpr int prologue_bias() { return LEAN ? 0 : 4; }
  // LDS operations younger than fragment G when step G waits for it
  static constexpr int younger(int G) {
    int cnt = 0;
    bool seen = false;
    for (int i = 0; i < PF; ++i) {
      if (seen) ++cnt;
      if (i == G) seen = true;
    }
    for (int s = 0; s < G; ++s) {
      if (seen) cnt += bias_at(s);
      if (s + PF < NIT) {
        if (seen) ++cnt;
        if (s + PF == G) seen = true;
      }
      if (seen) cnt += writes_at(s);
    }
    return cnt;
  }
};

// FfnStream (ldm_pipes.h): NIT = KS + 1 + 2 NT2 queue items per 32-wide hidden chunk (KS W1 fragments, one pseudo item
// for the bias / ReLU / cast step, 2 NT2 W2 fragments), PF deep, CONTINUOUS across chunks: step IT issues item IT + PF,
// which for IT >= NIT - PF belongs to the next chunk.  At step SYNC = NIT - PF, after the barrier and BEFORE that step's
// fragment read, the next chunk's bias is read (4 x ds_read_b128); it is the C operand of the next chunk's first MFMA.
template <int KS, int NT2, int PFQ>
struct FfnSched {
  static constexpr int PF = PFQ, NIT = KS + 1 + 2 * NT2, SYNC = NIT - PF;
  static_assert(NIT % PF == 0, "queue slots must line up across chunks");
  static_assert(PF - 1 + 4 <= 15, "lgkmcnt is a 4-bit counter");
  // LDS operations younger than item IT when step IT waits for it: PF - 1 fragments, + the 4 bias reads issued at SYNC
  // for the items that were issued before them (SYNC < IT < NIT: items IT issued at step IT - PF < SYNC ... see replay)
  static constexpr int after(int IT) { return IT > SYNC ? PF - 1 + 4 : PF - 1; }
};

}  // namespace ldm_sched
