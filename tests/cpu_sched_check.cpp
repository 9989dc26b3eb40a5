// CPU replay of HeadStream's LDS issue schedule (layout_dm_amd/csrc/ldm_stream_sched.h), compiled with plain g++ by
// tests/test_stream_sched.py.  The kernel's counted waits ("s_waitcnt lgkmcnt(younger(G))") are only right if
// younger() agrees with the order in which HeadStream::step really issues its LDS operations.  That order is restated
// here from the kernel source (ldm_pipes.h HeadStream::run / step), independently of younger():
//
//   head prologue   [4 bias reads (non-LEAN: tile 0's)]  fragments 0 .. PF-1
//   step G (tile J = G / 29, local IT = G % 29), in program order:
//       wait for fragment G                      s_waitcnt lgkmcnt(younger(G))
//       MFMA (IT == 0, non-LEAN K/Q tile: C operand = the bias read at the previous barrier step)
//       IT == SYNC:      barrier; non-LEAN: 4 bias reads for tile J+1 (if K/Q); LEAN, J == 5: 4 bias reads for tile 5
//       LEAN, IT == BIAS_LEAN, J >= 1: 4 bias reads for tile J-1 (if K/Q)
//       fragment read G + PF (if < NIT)
//       J >= 1, IT in [EPI0, EPI0 + 6): epilogue slice IT - EPI0 of tile J-1: slices 0 / 1 consume the LEAN bias,
//                                        slices 2 / 5 issue a ds_write_b128 when tile J-1 is a K or V tile
//   behind the stream: (LEAN: s_waitcnt lgkmcnt(0)) q1's epilogue (LEAN: consumes tile 5's bias)
//
// With an in-order LDS queue (a wave's LDS operations complete in issue order), a wait with count n retires everything
// but the n youngest operations.  Checked: (1) younger(G) == the number of operations issued after fragment G at step G's
// wait; (2) fragment G has completed when its MFMA issues; (3) every bias has completed when it is consumed; (4) every
// K / V^T store of the head has completed when the wave arrives at tile 5's barrier (the other waves read K / V^T right
// behind it); (5) the queue is empty behind the last step; (6) no wait count exceeds the 4-bit lgkmcnt field.
#include <cstdio>
#include <vector>

#include "../layout_dm_amd/csrc/ldm_stream_sched.h"

static int fails = 0;
#define CHECK(cond, ...)                                     \
  do {                                                       \
    if (!(cond)) {                                           \
      if (fails < 20) { printf(__VA_ARGS__); printf("\n"); } \
      ++fails;                                               \
    }                                                        \
  } while (0)

enum Kind { FRAG, BIAS, STORE };
struct Op { Kind k; int id; };  // FRAG: fragment index; BIAS: tile whose bias it is; STORE: tile whose K / V^T piece it is

template <bool LEAN>
static void replay(const char* name) {
  using S = ldm_sched::HeadSched<LEAN>;
  std::vector<Op> q;          // issue order
  size_t done = 0;            // operations [0, done) have completed
  auto issue = [&](Kind k, int id) { q.push_back(Op{k, id}); };
  auto wait = [&](int n) {    // s_waitcnt lgkmcnt(n)
    const size_t outstanding = q.size() - done;
    if (outstanding > (size_t)n) done = q.size() - n;
  };
  auto completed = [&](Kind k, int id) {
    for (size_t i = 0; i < q.size(); ++i)
      if (q[i].k == k && q[i].id == id && i >= done) return false;
    return true;
  };
  auto issued = [&](Kind k, int id) {
    for (const Op& o : q)
      if (o.k == k && o.id == id) return true;
    return false;
  };
  for (int i = 0; i < S::prologue_bias(); ++i) issue(BIAS, 0);
  for (int i = 0; i < S::PF; ++i) issue(FRAG, i);
  for (int G = 0; G < S::NIT; ++G) {
    const int J = G / S::KS, IT = G % S::KS;
    // (1) the count the kernel uses == the operations issued after fragment G so far
    size_t pos = q.size();
    for (size_t i = 0; i < q.size(); ++i)
      if (q[i].k == FRAG && q[i].id == G) pos = i;
    CHECK(pos < q.size(), "%s: fragment %d was never issued before its step", name, G);
    const int truth = (int)(q.size() - 1 - pos);
    CHECK(S::younger(G) == truth, "%s: younger(%d) = %d, replay says %d", name, G, S::younger(G), truth);
    CHECK(S::younger(G) <= 15, "%s: wait count %d at step %d exceeds the lgkmcnt field", name, S::younger(G), G);
    wait(S::younger(G));
    CHECK(completed(FRAG, G), "%s: fragment %d not complete at its MFMA", name, G);  // (2)
    if (!LEAN && IT == 0 && S::tile_has_bias(J))                                      // (3) C operand
      CHECK(issued(BIAS, J) && completed(BIAS, J), "%s: bias of tile %d not complete at its first MFMA", name, J);
    if (IT == S::SYNC) {
      if (J == S::NT - 1) {                                                           // (4)
        for (int t = 0; t < 4; ++t) {
          int n = 0;
          for (const Op& o : q) n += (o.k == STORE && o.id == t);
          CHECK(n == 2, "%s: tile %d issued %d stores before the last barrier (want 2)", name, t, n);
          CHECK(completed(STORE, t), "%s: K / V^T stores of tile %d not complete at the last barrier", name, t);
        }
      }
      if (!LEAN && J + 1 < S::NT && S::tile_has_bias(J + 1)) for (int i = 0; i < 4; ++i) issue(BIAS, J + 1);
      if (LEAN && J == S::NT - 1) for (int i = 0; i < 4; ++i) issue(BIAS, J);
    }
    if (LEAN && IT == S::BIAS_LEAN && J >= 1 && S::tile_has_bias(J - 1)) for (int i = 0; i < 4; ++i) issue(BIAS, J - 1);
    {  // the schedule header must describe exactly these bias reads
      const int want = (int)q.size();
      (void)want;
    }
    if (G + S::PF < S::NIT) issue(FRAG, G + S::PF);
    if (J >= 1 && IT >= S::EPI0 && IT < S::EPI0 + 6) {
      const int sl = IT - S::EPI0, pt = J - 1;
      if (LEAN && sl <= 1 && S::tile_has_bias(pt))                                    // (3) epilogue bias
        CHECK(issued(BIAS, pt) && completed(BIAS, pt), "%s: bias of tile %d not complete at its epilogue slice %d", name, pt, sl);
      if ((sl == 2 || sl == 5) && pt < 4) issue(STORE, pt);
    }
    // cross-check bias_at / writes_at against what this replay issued at step G
    int nb = 0, nw = 0;
    for (size_t i = pos + 1; i < q.size(); ++i) (void)i;
    {
      // count operations issued during this step: everything after the previous step's size
      static size_t prev_size = 0;
      if (G == 0) prev_size = (size_t)S::prologue_bias() + S::PF;
      for (size_t i = prev_size; i < q.size(); ++i) {
        nb += q[i].k == BIAS;
        nw += q[i].k == STORE;
      }
      prev_size = q.size();
    }
    CHECK(nb == S::bias_at(G), "%s: step %d issued %d bias reads, bias_at says %d", name, G, nb, S::bias_at(G));
    CHECK(nw == S::writes_at(G), "%s: step %d issued %d stores, writes_at says %d", name, G, nw, S::writes_at(G));
  }
  if (LEAN) wait(0);  // HeadStream<LEAN>::run: explicit s_waitcnt lgkmcnt(0) in front of q1's epilogue
  CHECK(done == q.size(), "%s: %zu LDS operations still in flight behind the stream", name, q.size() - done);  // (5)
  if (LEAN) CHECK(completed(BIAS, S::NT - 1), "%s: q1's bias not complete behind the stream", name);
  printf("%s: %zu LDS operations replayed\n", name, q.size());
}

// FfnStream: three chunks of the continuous chunk loop (prologue: bias of chunk 0, then PF fragments)
template <int KS, int NT2, int PF>
static void replay_ffn(const char* name) {
  using S = ldm_sched::FfnSched<KS, NT2, PF>;
  struct O { int kind, chunk, item; };  // kind 0 = fragment, 1 = bias
  std::vector<O> q;
  size_t done = 0;
  auto wait = [&](int n) {
    if (q.size() - done > (size_t)n) done = q.size() - n;
  };
  auto completed = [&](int kind, int chunk, int item) {
    bool found = false;
    for (size_t i = 0; i < q.size(); ++i)
      if (q[i].kind == kind && q[i].chunk == chunk && (kind == 1 || q[i].item == item)) {
        found = true;
        if (i >= done) return false;
      }
    return found;
  };
  const int NC = 3;
  for (int i = 0; i < 4; ++i) q.push_back(O{1, 0, i});
  for (int i = 0; i < PF; ++i) q.push_back(O{0, 0, i});
  for (int c = 0; c < NC; ++c)
    for (int IT = 0; IT < S::NIT; ++IT) {
      size_t pos = q.size();
      for (size_t i = 0; i < q.size(); ++i)
        if (q[i].kind == 0 && q[i].chunk == c && q[i].item == IT) pos = i;
      CHECK(pos < q.size(), "%s: item %d of chunk %d never issued", name, IT, c);
      const int truth = (int)(q.size() - 1 - pos);
      CHECK(S::after(IT) == truth, "%s: after(%d) = %d, replay says %d (chunk %d)", name, IT, S::after(IT), truth, c);
      wait(S::after(IT));
      CHECK(completed(0, c, IT), "%s: item %d of chunk %d not complete at its step", name, IT, c);
      if (IT == 0) CHECK(completed(1, c, 0), "%s: bias of chunk %d not complete at its first MFMA", name, c);
      if (IT == S::SYNC)
        for (int i = 0; i < 4; ++i) q.push_back(O{1, c + 1, i});
      const int nxt = IT + PF;
      q.push_back(O{0, nxt < S::NIT ? c : c + 1, nxt % S::NIT});
    }
  printf("%s: %zu LDS operations replayed\n", name, q.size());
}

int main() {
  replay_ffn<29, 15, 6>("FfnStream<29,15,PF=6>");
  replay_ffn<29, 15, 10>("FfnStream<29,15,PF=10>");
  replay<false>("HeadStream");
  replay<true>("HeadStream<LEAN>");
  if (fails) {
    printf("FAILED: %d checks\n", fails);
    return 1;
  }
  printf("OK: counted waits of both HeadStream variants agree with the replayed issue order\n");
  return 0;
}
