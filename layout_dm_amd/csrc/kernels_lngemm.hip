// Row-resident (Ada)LayerNorm + fp16 x 3 GEMM for the split (reference-precision) numerics mode: ONE launch where the
// per-step path had two (a LayerNorm row kernel writing hi / lo operand rows to HBM, then gemm16x3_k reading them back
// tile by tile).
//
//   out[M, N] = epi( LN(x)[M, 464] * W[N, 464]^T )        x rows in, LN output never leaves the registers
//
// Used for the three LayerNorm-fed GEMMs of a denoiser pass (transformer_utils.py:165-246, nn_lib.py:186-237):
//   AdaLN  + in_proj (QKV)      fp32 out (the fp16 x 3 attention splits its own operands), also writes AdaLN(x): the block's
//                               residual base (Block.forward: residual on the NORMED x)
//   norm2  + linear1 + ReLU     hi / lo fp16 out (the A operand of gemm16x3_k for linear2)
//   head LN + vocabulary head   fp32 logits
// and, r05 second half, optionally a GEMM in FRONT of the LayerNorm as well (PRE, "GEMM prologue", further down): the launch
// computes its rows x = residual + bias + scale * (A . Wpre^T) instead of reading them.  Shipped for linear2 — in front of the
// next layer's AdaLN + in_proj and of the head — so that a layer is three launches besides the attention:
//   [linear2 of layer l - 1 +] AdaLN + in_proj   |   attention   |   out_proj (gemm16x3_k)   |   norm2 + linear1 + ReLU
//
// Why: gemm16x3_k is bound by its operand fills (one 64-KiB stage in flight per ~1.7 us of fill latency), and a
// 256 x 256 tile pulls FOUR images per stage (A hi, A lo, W hi, W lo).  Here a workgroup owns 128 rows for the whole GEMM —
// the structure of the fast mode's stack kernel (kernels_stack.hip): their normalised hi / lo MFMA fragments sit in
// registers (29 k16-steps x 2 x 4 registers per wave: hi in arch VGPRs, lo parked in AGPRs), only the WEIGHTS stream —
// hi | lo tile images of 32 output columns, 64 KiB per stage, linear LDS-DMA through a 2-stage ring — i.e. half the fill
// bytes per flop, no A operand traffic at all, no LayerNorm launch and no hi / lo activation round trip through HBM.
// Per k16-step: two ds_read_b128 (W hi, W lo fragment) and three MFMAs (W_hi x_hi + W_hi x_lo + W_lo x_hi into ONE sum,
// lo unscaled: ldm_kernels.h kSplitLoScale); the read queue is continuous across tiles (FfnStream's protocol: one
// s_waitcnt vmcnt(0) + s_barrier per tile at step NIT - PF); a tile runs on two accumulator chains and its epilogue — sum,
// transpose through LDS so that every row segment leaves as whole 64 / 128-byte pieces, scale, bias, ReLU, hi / lo split — is
// issued in small branch-free slices beside the MFMAs of tile t + 1.
// LayerNorm arithmetic: two-pass (mean, then sum of squared deviations) in fp32 on the row's registers, eps 1e-5, like
// ln_rows (kernels_norm.hip).  Weight K axis in MFMA k-slot order (ldm_pack::kslot): a lane's accumulator-layout registers
// of column groups 2ks, 2ks + 1 ARE its B fragment of k16-step ks.
// gfx950 only; geometry: D == 464 (29 k16-steps), N % 4 == 0, N <= 2048.
#include "ldm_dma.h"
#include "ldm_kernels.h"
#include "ldm_pipes.h"

namespace ldm {

namespace {

// A tile is NIT = 30 queue items: the 29 k16-steps and ONE pseudo item (nothing is read, nothing is issued) so that
// NIT % PF == 0 — item i of every tile then lives in queue slot i % PF and the queue runs on across tiles (FfnStream's trick;
// r05's first GPU run had 29 items on the 6-deep queue: item 0 of the next tile landed in the slot of an unconsumed pair).
// PF = 6 fragment pairs in flight.  Measured alternatives that did NOT ship (same-box A/B builds,
// profiles/r05_call7_lngemm_queue_depth_and_agpr_accumulators_ab.txt): PF = 8 (32 items per tile) leaves 14 fragment reads
// behind the awaited pair, the epilogue's LDS operations come on top and the 4-bit lgkmcnt counter no longer covers what is in
// flight — NaN logits; the tile accumulators in AGPRs ("+a" MFMA destinations beside the AGPR-resident lo fragments) lose low-order
// terms (logits error 5e-5 instead of 9e-7) and are not faster.
constexpr int LG_KS = 29, LG_NIT = 30, LG_PF = 6, LG_SYNC = LG_NIT - LG_PF;
static_assert(LG_NIT % LG_PF == 0, "queue slots must line up across tiles");
constexpr bool lg_real(int i) { return (i % LG_NIT) < LG_KS; }
// LDS reads issued behind the pair of item IT when step IT waits for it: the real items among IT + 1 .. IT + PF - 1
// (rpi = LDS reads per item: W hi and W lo, or — W2, the two-product form: weights fp16 only — W hi alone)
constexpr int lg_younger(int IT, int rpi = 2) {
  int n = 0;
  for (int j = 1; j < LG_PF; ++j) n += lg_real(IT + j) ? rpi : 0;
  return n;
}
static_assert(lg_younger(0) <= 15, "lgkmcnt is a 4-bit counter");
constexpr int LG_STAGE = 65536;                    // W hi tile (32 KiB) | W lo tile (32 KiB)
constexpr int LG_LO = 32768;
constexpr int LG_TP_LD = 36;                       // floats per row of a wave's 32 x 32 transpose buffer (16-B aligned rows)
constexpr int LG_TP_BYTES = 32 * LG_TP_LD * 4;     // 4 608 B per wave
constexpr int LG_PAR_OFF = 2 * LG_STAGE;           // multiplier [512] | shift [512]
constexpr int LG_BIAS_OFF = LG_PAR_OFF + 2 * 512 * 4;   // bias [2048]
constexpr int LG_TP_OFF = LG_BIAS_OFF + 2048 * 4;
constexpr int LG_PBIAS_OFF = LG_TP_OFF + 4 * LG_TP_BYTES;   // bias of the GEMM prologue [512]
constexpr int LG_LDS = LG_PBIAS_OFF + 512 * 4;              // 163 840 B = all of a CU's LDS
static_assert(LG_LDS <= 160 * 1024, "LDS budget");

template <int OFF>
__device__ __forceinline__ void lg_dsr(f16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}

__device__ unsigned long long g_lngemm_phase[8];   // TM instantiation: workgroups | total | prologue | sync waits | lgkm waits | tail

struct LgState {
  unsigned long long t_sync = 0, t_lgkm = 0;   // (TM) cycles inside the per-tile vmcnt + barrier, inside the counted LDS waits
  f16x8 qh[LG_PF], ql[LG_PF];   // W hi / lo fragment queue
  unsigned aW[8];               // LDS byte addresses of the fragment columns in the CURRENT stage
  const f16x8* xhi;             // [29] activation fragments, hi (arch VGPRs)
  const f16x8* xlo;             // [29] ... lo (AGPRs)
  f32x16 accA, accB;            // the tile's TWO accumulator chains: the 87 MFMAs of a tile alternate between them (summed in the epilogue)
  const char* img;              // weight image + wave * 16 KiB
  unsigned lds_w;               // lds0 + wave * 16 KiB
  unsigned voff;                // lane * 16
  int stage_delta;              // +- 64 KiB: what moves aW from the current tile's stage to the next tile's
  int n_tiles;
  // the tile whose DMA is in progress: global / LDS address of its NEXT 4-KiB group of this wave's 16 KiB (uniform)
  const char* dma_g;
  unsigned dma_l;
};

// item IT of the tile whose stage aW points at: W hi and W lo fragment of k16-step IT (pseudo items: nothing)
template <int IT, bool W2 = false>
__device__ __forceinline__ void lg_read(LgState& s) {
  if constexpr (IT < LG_KS) {
    lg_dsr<256 * (IT >> 3)>(s.qh[IT % LG_PF], s.aW[IT & 7]);
    if constexpr (!W2) lg_dsr<256 * (IT >> 3) + LG_LO>(s.ql[IT % LG_PF], s.aW[IT & 7]);
  }
}

// the kernel prologue primes the queue: items 0 .. PF - 1 of tile 0
template <int I, bool W2 = false>
__device__ __forceinline__ void lg_prime(LgState& s) {
  if constexpr (I < LG_PF) {
    lg_read<I, W2>(s);
    lg_prime<I + 1, W2>(s);
  }
}

// Start the DMA of tile td (clamped to the last tile: re-loading it into a free stage is harmless and keeps the stream free of
// branches) into stage td & 1: piece J = 1 KiB of this wave's 16; the pieces of one tile straddle a tile boundary.
__device__ __forceinline__ void lg_dma_begin(LgState& s, int td) {
  const int t = td < s.n_tiles ? td : s.n_tiles - 1;
  s.dma_g = s.img + (size_t)t * LG_STAGE;
  s.dma_l = s.lds_w + (unsigned)(td & 1) * LG_STAGE;
}
typedef float lg_f32x4 __attribute__((ext_vector_type(4)));   // (a 4-register asm operand: HIP's float4 is a struct)
typedef unsigned lg_u32x2 __attribute__((ext_vector_type(2)));
struct LgEpi {
  unsigned a_bias;              // LDS byte address of the bias table + (lane & 7) * 16: the lane's 4 columns AFTER the transpose
  unsigned a_tpw, a_tpr;        // this lane's write / read address in the wave's transpose buffer (LDS bytes)
  const char *C0, *C1;          // output bases: fp32 C | fp16 hi, fp16 lo
  const char *b0, *b1;          // ... of the tile whose epilogue is in flight (uniform: + tile * 32 columns)
  size_t tstride;               // bytes per tile in C0 / C1: 128 (fp32 rows) | 64 (fp16 rows) | one panel (OUT = 2: panel-major hi / lo)
  unsigned voff[4];             // this lane's byte offset in 8-row pass p: (row0 + 8 p + lane / 8) * ld + (lane & 7) * 4 columns
  unsigned long long rowmask[4];   // lanes whose row of pass p exists (row < M)
  unsigned long long colmask;   // lanes whose 4 columns of the tile in flight exist (col < N; N % 4 == 0)
  int N, c4;                    // c4 = (lane & 7) * 4
  float out_scale;
  lg_f32x4 bb;                  // bias of the lane's 4 columns of the tile whose epilogue is in flight
  lg_f32x4 ev[4];               // the tile transposed: 8 lanes per row
};

// The epilogue of a tile.  Its first part is exposed, at the tile's pseudo step: the two accumulator chains are summed and go to
// the wave's transpose buffer (accumulator layout -> row-major 32 x 32) — the next tile's first MFMAs re-initialise both chains.
// The rest runs in small slices beside the MFMAs of the NEXT tile.  Every LDS operation is issued through asm: hipcc's
// own s_waitcnt for a load it knows about would be lgkmcnt(0), i.e. a drain of the whole fragment queue.  What guarantees that a
// slice's data has landed is the per-step counted wait: an operation issued at step s is older than the fragment pairs of items
// s + PF .. and therefore complete once step s + PF has waited (in-order LDS completion).
//   step KS (pseudo)   chain A + chain B, 4 x ds_write_b128                                          (exposed: ~200 cycles per tile)
//   step 1             the bias of the lane's 4 columns in the transposed view (1 x ds_read_b128)      -> landed at step 7
//   steps 2, 3         the tile back, 8 lanes per row (2 x ds_read_b128 each)                          -> landed at step 9
//   steps SYNC + 1 .. SYNC + 4   one 8-row pass each: scale + bias, (ReLU, hi / lo split,) 128- / 64-byte row segments to global memory
//                      (behind the barrier: the next s_waitcnt vmcnt(0), which cannot tell stores from DMA pieces, is a whole
//                      tile away)
// At most 2 extra LDS operations per step: 10 (counted wait) + 2 (the step's own pair) + 2 = 14 in flight (lgkmcnt: 4 bits).
// A store pass is ONE branch-free asm statement (r05 call 17: every non-MFMA instruction of this loop costs ~6 cycles of issue,
// hidden or not; hipcc's version of a pass was 62 - 75 instructions with eight branches): rows beyond M and columns beyond N are
// masked through EXEC (SGPR lane masks, restored inside the statement), the address is an SGPR base + a per-lane 32-bit offset that
// does not change from tile to tile.  OUT = 0: fp32; OUT = 1: ReLU, then hi = fp16(v) and lo = fp16(v - hi) (kSplitLoScale == 1).
static_assert(kSplitLoScale == 1.0f, "the lo half is stored unscaled");
// chain A + chain B -> the wave's transpose buffer (the exposed part)
__device__ __forceinline__ void lg_epilogue_sum_write(const LgEpi& e, f32x16& a, f32x16& b) {
  wait_lgkm<8>();   // (4 ds_write_b128 follow: lgkmcnt is a 4-bit counter)
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a), "+v"(b));   // the chains' last MFMAs (8 passes each): results visible to the VALU
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const lg_f32x4 v = {a[rq * 4 + 0] + b[rq * 4 + 0], a[rq * 4 + 1] + b[rq * 4 + 1], a[rq * 4 + 2] + b[rq * 4 + 2],
                        a[rq * 4 + 3] + b[rq * 4 + 3]};
    if (rq == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(e.a_tpw), "v"(v) : "memory");
    if (rq == 1) asm volatile("ds_write_b128 %0, %1 offset:32" ::"v"(e.a_tpw), "v"(v) : "memory");
    if (rq == 2) asm volatile("ds_write_b128 %0, %1 offset:64" ::"v"(e.a_tpw), "v"(v) : "memory");
    if (rq == 3) asm volatile("ds_write_b128 %0, %1 offset:96" ::"v"(e.a_tpw), "v"(v) : "memory");
  }
}
template <int IT, int OUT, int ABL = 0>
__device__ __forceinline__ void lg_epilogue_slice(LgEpi& e, int tile) {
  if constexpr (IT == 1) {
    const unsigned ab = e.a_bias + (unsigned)tile * 128;
    asm volatile("ds_read_b128 %0, %1" : "=v"(e.bb) : "v"(ab) : "memory");
    // (tile == -1, the slices of the first tile's steps: an unsigned compare masks every lane — the passes run and store nothing)
    e.colmask = __ballot((unsigned)(tile * 32 + e.c4) < (unsigned)e.N);
    // (measurement variant 32: every tile stores to the columns of tile 0 — the same bytes per store, an L2-resident target)
    e.b0 = e.C0 + ((ABL & 32) ? 0 : (size_t)tile * e.tstride);
    e.b1 = e.C1 + ((ABL & 32) ? 0 : (size_t)tile * e.tstride);
  }
  if constexpr (IT == 2 || IT == 3) {
    constexpr int p0 = (IT - 2) * 2;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(e.ev[p0]) : "v"(e.a_tpr), "n"(p0 * 8 * LG_TP_LD * 4) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(e.ev[p0 + 1]) : "v"(e.a_tpr), "n"((p0 + 1) * 8 * LG_TP_LD * 4) : "memory");
  }
  if constexpr (IT > LG_SYNC && IT <= LG_SYNC + 4) {
    constexpr int p = IT - LG_SYNC - 1;
    const lg_f32x4 ev = e.ev[p];
    float t0, t1, t2, t3;
    if constexpr (OUT == 0) {
      asm volatile(
          "v_fma_f32 %[t0], %[e0], %[sc], %[b0]\n\tv_fma_f32 %[t1], %[e1], %[sc], %[b1]\n\t"
          "v_fma_f32 %[t2], %[e2], %[sc], %[b2]\n\tv_fma_f32 %[t3], %[e3], %[sc], %[b3]"
          : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
          : [e0] "v"(ev.x), [e1] "v"(ev.y), [e2] "v"(ev.z), [e3] "v"(ev.w), [sc] "s"(e.out_scale), [b0] "v"(e.bb.x), [b1] "v"(e.bb.y),
            [b2] "v"(e.bb.z), [b3] "v"(e.bb.w));
      const lg_f32x4 T = {t0, t1, t2, t3};
      if constexpr (ABL & 16) asm volatile("" ::"v"(T));   // (measurement variant 16: the pass without its store)
      else if constexpr (ABL & 64)   // (measurement variant 64: non-temporal stores)
        asm volatile("s_and_b64 exec, %[rm], %[cm]\n\tglobal_store_dwordx4 %[vo], %[T], %[b] nt\n\ts_mov_b64 exec, -1"
                     ::[rm] "s"(e.rowmask[p]), [cm] "s"(e.colmask), [vo] "v"(e.voff[p]), [T] "v"(T), [b] "s"(e.b0)
                     : "memory");
      else asm volatile("s_and_b64 exec, %[rm], %[cm]\n\tglobal_store_dwordx4 %[vo], %[T], %[b]\n\ts_mov_b64 exec, -1"
                   ::[rm] "s"(e.rowmask[p]), [cm] "s"(e.colmask), [vo] "v"(e.voff[p]), [T] "v"(T), [b] "s"(e.b0)
                   : "memory");
    } else if constexpr (OUT == 3) {   // ReLU, fp16 — no lo half (plain-fp16 hidden activations: the hybrid mode's linear1)
      unsigned h01, h23;
      asm volatile("v_fma_f32 %[t0], %[e0], %[sc], %[b0]\n\tv_fma_f32 %[t1], %[e1], %[sc], %[b1]\n\t"
                   "v_fma_f32 %[t2], %[e2], %[sc], %[b2]\n\tv_fma_f32 %[t3], %[e3], %[sc], %[b3]\n\t"
                   "v_max_f32 %[t0], 0, %[t0]\n\tv_max_f32 %[t1], 0, %[t1]\n\tv_max_f32 %[t2], 0, %[t2]\n\tv_max_f32 %[t3], 0, %[t3]\n\t"
                   "v_cvt_pk_f16_f32 %[h01], %[t0], %[t1]\n\tv_cvt_pk_f16_f32 %[h23], %[t2], %[t3]"
                   : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [h01] "=&v"(h01), [h23] "=&v"(h23)
                   : [e0] "v"(ev.x), [e1] "v"(ev.y), [e2] "v"(ev.z), [e3] "v"(ev.w), [sc] "s"(e.out_scale), [b0] "v"(e.bb.x), [b1] "v"(e.bb.y),
                     [b2] "v"(e.bb.z), [b3] "v"(e.bb.w));
      const lg_u32x2 H = {h01, h23};
      asm volatile("s_and_b64 exec, %[rm], %[cm]\n\tglobal_store_dwordx2 %[vo], %[H], %[bh]\n\ts_mov_b64 exec, -1"
                   ::[rm] "s"(e.rowmask[p]), [cm] "s"(e.colmask), [vo] "v"(e.voff[p]), [H] "v"(H), [bh] "s"(e.b0)
                   : "memory");
    } else {
      unsigned h01, h23, l01, l23;
      // scale + bias, (ReLU,) hi = fp16(v), lo = fp16(v - float(hi)): exact in fp32 (one fma_mix per value: -hi(f16 half of the pair) * 1.0 + v)
#define LG_SPLIT_PASS(RELU4)                                                                                                             \
  asm volatile("v_fma_f32 %[t0], %[e0], %[sc], %[b0]\n\tv_fma_f32 %[t1], %[e1], %[sc], %[b1]\n\t"                                         \
               "v_fma_f32 %[t2], %[e2], %[sc], %[b2]\n\tv_fma_f32 %[t3], %[e3], %[sc], %[b3]\n\t" RELU4                                   \
               "v_cvt_pk_f16_f32 %[h01], %[t0], %[t1]\n\tv_cvt_pk_f16_f32 %[h23], %[t2], %[t3]\n\t"                                       \
               "v_fma_mix_f32 %[t0], -%[h01], 1.0, %[t0] op_sel_hi:[1,0,0]\n\t"                                                          \
               "v_fma_mix_f32 %[t1], -%[h01], 1.0, %[t1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                                           \
               "v_fma_mix_f32 %[t2], -%[h23], 1.0, %[t2] op_sel_hi:[1,0,0]\n\t"                                                          \
               "v_fma_mix_f32 %[t3], -%[h23], 1.0, %[t3] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                                           \
               "v_cvt_pk_f16_f32 %[l01], %[t0], %[t1]\n\tv_cvt_pk_f16_f32 %[l23], %[t2], %[t3]"                                           \
               : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [h01] "=&v"(h01), [h23] "=&v"(h23), [l01] "=&v"(l01),     \
                 [l23] "=&v"(l23)                                                                                                         \
               : [e0] "v"(ev.x), [e1] "v"(ev.y), [e2] "v"(ev.z), [e3] "v"(ev.w), [sc] "s"(e.out_scale), [b0] "v"(e.bb.x), [b1] "v"(e.bb.y), \
                 [b2] "v"(e.bb.z), [b3] "v"(e.bb.w))
      if constexpr (OUT == 1)
        LG_SPLIT_PASS("v_max_f32 %[t0], 0, %[t0]\n\tv_max_f32 %[t1], 0, %[t1]\n\tv_max_f32 %[t2], 0, %[t2]\n\tv_max_f32 %[t3], 0, %[t3]\n\t");
      else   // OUT == 2: q / k / v, no ReLU
        LG_SPLIT_PASS("");
#undef LG_SPLIT_PASS
      const lg_u32x2 H = {h01, h23}, L = {l01, l23};
      if constexpr (ABL & 16) asm volatile("" ::"v"(H), "v"(L));
      else if constexpr (ABL & 64)
        asm volatile("s_and_b64 exec, %[rm], %[cm]\n\tglobal_store_dwordx2 %[vo], %[H], %[bh] nt\n\tglobal_store_dwordx2 %[vo], %[L], %[bl] nt\n\t"
                     "s_mov_b64 exec, -1"
                     ::[rm] "s"(e.rowmask[p]), [cm] "s"(e.colmask), [vo] "v"(e.voff[p]), [H] "v"(H), [L] "v"(L), [bh] "s"(e.b0),
                     [bl] "s"(e.b1)
                     : "memory");
      else asm volatile("s_and_b64 exec, %[rm], %[cm]\n\tglobal_store_dwordx2 %[vo], %[H], %[bh]\n\tglobal_store_dwordx2 %[vo], %[L], %[bl]\n\t"
                   "s_mov_b64 exec, -1"
                   ::[rm] "s"(e.rowmask[p]), [cm] "s"(e.colmask), [vo] "v"(e.voff[p]), [H] "v"(H), [L] "v"(L), [bh] "s"(e.b0),
                   [bl] "s"(e.b1)
                   : "memory");
    }
  }
}
constexpr bool lg_slice_step(int IT) {
  return IT == 1 || IT == 2 || IT == 3 || (IT > LG_SYNC && IT <= LG_SYNC + 4);
}

// One tile = NIT steps.  A real step (a k16-step) is ONE asm statement: the counted wait, then its three MFMAs with the step's
// LDS-DMA piece and its two fragment reads BETWEEN them:
//     s_waitcnt lgkmcnt(n)        ; fragment pair of this item landed
//     [s_mov_b32 m0, <LDS group>] ; (steps that carry a DMA piece)
//     v_mfma  c0 += Wh x_hi
//     [global_load_lds_dwordx4]   ; the piece: 1 KiB of weight tile t + 1 / t + 2
//     v_mfma  c1 += Wh x_lo
//     ds_read_b128 Wh'            ; the pair PF items ahead, into the queue slot this step is consuming: Wh is dead from here
//     v_mfma  c0 += Wl x_hi
//     ds_read_b128 Wl'
// c0 / c1 are the tile's two accumulator chains, swapped every step: consecutive MFMAs are ALWAYS on different accumulators
// (A B A | B A B | ...), each chain gets 3 MFMAs per two steps, and the two are summed in the epilogue (lo unscaled: any chain may
// take any of the three products).
// What r05 measured on the way here (profiles/r05_call16_17_lngemm_compile_time_variants.txt, r05_call18_21_lngemm_store_cost.txt;
// linear1, 58 tiles x 87 MFMAs per wave, 128 rows per workgroup, us per launch): prologue + empty loop 36; the MFMAs alone +90
// (the matrix-pipe floor at the ~1.8 - 1.9 GHz a dense MFMA stream clocks at: chain arrangement — one chain, two alternating per
// MFMA or per step — does not move it); fragment reads +11, DMA pieces +10, epilogue +44.  With one wave per SIMD the parts ADD:
// a filler between two MFMAs costs about what it costs behind them (~6 cycles of issue per non-MFMA instruction,
// MI355X_MICROARCH.md's issue-slot table), so what pays is instruction COUNT — this form has ~260 non-MFMA instructions per tile
// and wave where the first had ~450 (hipcc's epilogue passes: 62 - 75 instructions and eight branches each) — and, for the
// epilogue, bytes: 33 of linear1's last 44 us are its 237 MB of hi / lo hidden rows on their way to HBM (6 us with the same stores
// aimed at an L2-resident target).  207 -> 187 us per launch over these steps; in_proj 166 -> 156.
// The counted waits are never larger than the number of LDS operations really issued behind the awaited fragment pair (the
// slices' extra operations only make them stricter).  The reads run on into the NEXT tile's stage unconditionally: behind the
// last tile they fetch bytes nobody uses (the stage exists; keeps the stream free of branches and the waits uniform).
#define LG_MFMA(D, B, A, C) "v_mfma_f32_32x32x16_f16 " D ", " B ", " A ", " C "\n\t"
#define LG_STEP_ASM(C0, C1, M0SET, PIECE, RD_HI, RD_LO)                                                                        \
  asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" M0SET LG_MFMA("%[c0]", "%[qh]", "%[xh]", C0) PIECE                                 \
                   LG_MFMA("%[c1]", "%[qh]", "%[xl]", C1) RD_HI LG_MFMA("%[c0]", "%[ql]", "%[xh]", "%[c0]") RD_LO              \
               : [c0] "+v"(c0), [c1] "+v"(c1), [qh] "+v"(qh), [ql] "+v"(ql)                                                     \
               : [xh] "v"(s.xhi[IT]), [xl] "a"(s.xlo[IT]), [aw] "v"(aw), [vo] "v"(s.voff), [dg] "s"(s.dma_g), [dl] "s"(s.dma_l), \
                 [w] "n"(W), [ro] "n"(RO), [rl] "n"(RO + LG_LO), [doff] "n"(DOFF)                                               \
               : "memory")
// W2 (two products: W_hi x_hi + W_hi x_lo, the weights' lo half neither read nor multiplied): chain c0 takes the first, c1 the second — A B | A B
#define LG_STEP_ASM2(C0, C1, M0SET, PIECE, RD_HI)                                                                               \
  asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" M0SET LG_MFMA("%[c0]", "%[qh]", "%[xh]", C0) PIECE                                 \
                   LG_MFMA("%[c1]", "%[qh]", "%[xl]", C1) RD_HI                                                                 \
               : [c0] "+v"(c0), [c1] "+v"(c1), [qh] "+v"(qh)                                                                    \
               : [xh] "v"(s.xhi[IT]), [xl] "a"(s.xlo[IT]), [aw] "v"(aw), [vo] "v"(s.voff), [dg] "s"(s.dma_g), [dl] "s"(s.dma_l), \
                 [w] "n"(W), [ro] "n"(RO), [doff] "n"(DOFF)                                                                     \
               : "memory")
// NP = 1 (plain fp16: W_hi x_hi only — the x_lo fragments do not exist): the chains alternate by step
// (XC: the register file of the step's x fragment — the odd k16-steps' fragments are parked in AGPRs by the kernel's prologue, like the
//  x_lo fragments of the other forms: with all 116 fragment registers in arch VGPRs hipcc spilt the prologue's raw rows to scratch)
#define LG_STEP_ASM1X(XC, C0, M0SET, PIECE, RD_HI)                                                                              \
  asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" M0SET LG_MFMA("%[c0]", "%[qh]", "%[xh]", C0) PIECE RD_HI                           \
               : [c0] "+v"(c0), [qh] "+v"(qh)                                                                                    \
               : [xh] XC(s.xhi[IT]), [aw] "v"(aw), [vo] "v"(s.voff), [dg] "s"(s.dma_g), [dl] "s"(s.dma_l),                        \
                 [w] "n"(W), [ro] "n"(RO), [doff] "n"(DOFF)                                                                     \
               : "memory")
#define LG_STEP_ASM1(C0, M0SET, PIECE, RD_HI)                          \
  do {                                                                 \
    if constexpr (IT & 1) LG_STEP_ASM1X("a", C0, M0SET, PIECE, RD_HI); \
    else LG_STEP_ASM1X("v", C0, M0SET, PIECE, RD_HI);                  \
  } while (0)
#define LG_A_M0 "s_mov_b32 m0, %[dl]\n\t"
#define LG_A_PIECE "global_load_lds_dwordx4 %[vo], %[dg] offset:%[doff]\n\t"
#define LG_A_RDH "ds_read_b128 %[qh], %[aw] offset:%[ro]\n\t"
#define LG_A_RDL "ds_read_b128 %[ql], %[aw] offset:%[rl]"
#define LG_A_RDH2 "ds_read_b128 %[qh], %[aw] offset:%[ro]"

// which DMA piece (of the 16 per tile and wave) step IT carries, or -1: pieces 0 .. NIT - 2 - SYNC of tile + 2 at steps SYNC + 1 ..
// NIT - 1 (its stage — this tile's — is free behind this tile's barrier), the rest at the first steps of the next tile
constexpr int lg_piece(int IT) {
  if (IT > LG_SYNC) return IT - LG_SYNC - 1;
  if (IT + (LG_NIT - 1 - LG_SYNC) < 16) return IT + (LG_NIT - 1 - LG_SYNC);
  return -1;
}
static_assert(lg_piece(LG_SYNC) == -1 && lg_piece(LG_SYNC + 1) == 0 && lg_piece(0) == lg_piece(LG_NIT - 1) + 1, "16 pieces, in order, none at the barrier step");

template <int IT, int OUT, bool TM = false, int ABL = 0, int NP = 3>   // NP products per k16-step: 3 | 2 (weights fp16 only) | 1 (plain fp16: x_hi too)
__device__ __forceinline__ void lg_step(LgState& s, LgEpi& e, int tile) {
  constexpr bool W2 = NP < 3;
  // ABL (measurement builds only, LDM_LNGEMM_ABL): compile-time removal of 2 = the fragment reads and their counted waits,
  // 4 = the weight DMA, 8 = the epilogue (sum, transpose, stores), 16 = the epilogue's global stores only, 32 = every tile's stores
  // aimed at the columns of tile 0 — timing variants of this loop, results meaningless
  constexpr bool kRd = !(ABL & 2), kDm = !(ABL & 4), kEp = !(ABL & 8);
  if constexpr (IT < LG_NIT) {
    constexpr int J = lg_piece(IT);
    constexpr bool hasD = kDm && J >= 0 && (!W2 || J < 8);   // (W2: only the stage's hi half is fetched — 8 KiB per wave, pieces 0 .. 7)
    // the DMA stream's uniform address state: a new tile at piece 0, the next 4-KiB group of this wave's 16 KiB at pieces 4, 8, 12
    if constexpr (hasD && J == 0) lg_dma_begin(s, tile + 2);
    if constexpr (hasD && J > 0 && (J & 3) == 0) {
      s.dma_g += 4096;
      s.dma_l += 4096;
    }
    // the fragment pair PF items ahead (pseudo items: nothing); the one behind the barrier step is read behind the barrier
    constexpr int RI = (IT + LG_PF) % LG_NIT;
    constexpr bool hasR = kRd && IT != LG_SYNC && RI < LG_KS;
    constexpr int RO = 256 * (RI >> 3);
    constexpr int DOFF = hasD ? (J & 3) * 1024 : 0;
    if constexpr (IT < LG_KS) {
      // LDS operations of a wave complete in order: all but the lg_younger(IT) youngest = the fragment pairs issued behind item IT's
      constexpr int W = kRd ? lg_younger(IT, W2 ? 1 : 2) : 15;
      f32x16& c0 = ((IT & 1) && NP != 2) ? s.accB : s.accA;   // two of the step's MFMAs (NP = 2: one; NP = 1: the step's only one)
      f32x16& c1 = ((IT & 1) && NP != 2) ? s.accA : s.accB;   // one (NP = 1: unused)
      f16x8& qh = s.qh[IT % LG_PF];
      f16x8& ql = s.ql[IT % LG_PF];
      const unsigned aw = s.aW[RI & 7];
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NP == 1 && IT < 2) {   // one product: step 0 starts chain A, step 1 chain B
        if constexpr (hasD && hasR) LG_STEP_ASM1("0", LG_A_M0, LG_A_PIECE, LG_A_RDH2);
        else if constexpr (hasD) LG_STEP_ASM1("0", LG_A_M0, LG_A_PIECE, "");
        else if constexpr (hasR) LG_STEP_ASM1("0", "", "", LG_A_RDH2);
        else LG_STEP_ASM1("0", "", "", "");
      } else if constexpr (NP == 1) {
        if constexpr (hasD && hasR) LG_STEP_ASM1("%[c0]", LG_A_M0, LG_A_PIECE, LG_A_RDH2);
        else if constexpr (hasD) LG_STEP_ASM1("%[c0]", LG_A_M0, LG_A_PIECE, "");
        else if constexpr (hasR) LG_STEP_ASM1("%[c0]", "", "", LG_A_RDH2);
        else LG_STEP_ASM1("%[c0]", "", "", "");
      } else if constexpr (W2 && IT == 0) {
        if constexpr (hasD && hasR) LG_STEP_ASM2("0", "0", LG_A_M0, LG_A_PIECE, LG_A_RDH2);
        else if constexpr (hasD) LG_STEP_ASM2("0", "0", LG_A_M0, LG_A_PIECE, "");
        else if constexpr (hasR) LG_STEP_ASM2("0", "0", "", "", LG_A_RDH2);
        else LG_STEP_ASM2("0", "0", "", "", "");
      } else if constexpr (W2) {
        if constexpr (hasD && hasR) LG_STEP_ASM2("%[c0]", "%[c1]", LG_A_M0, LG_A_PIECE, LG_A_RDH2);
        else if constexpr (hasD) LG_STEP_ASM2("%[c0]", "%[c1]", LG_A_M0, LG_A_PIECE, "");
        else if constexpr (hasR) LG_STEP_ASM2("%[c0]", "%[c1]", "", "", LG_A_RDH2);
        else LG_STEP_ASM2("%[c0]", "%[c1]", "", "", "");
      } else if constexpr (IT == 0) {  // both chains start from zero
        if constexpr (hasD && hasR) LG_STEP_ASM("0", "0", LG_A_M0, LG_A_PIECE, LG_A_RDH, LG_A_RDL);
        else if constexpr (hasD) LG_STEP_ASM("0", "0", LG_A_M0, LG_A_PIECE, "", "");
        else if constexpr (hasR) LG_STEP_ASM("0", "0", "", "", LG_A_RDH, LG_A_RDL);
        else LG_STEP_ASM("0", "0", "", "", "", "");
      } else {
        if constexpr (hasD && hasR) LG_STEP_ASM("%[c0]", "%[c1]", LG_A_M0, LG_A_PIECE, LG_A_RDH, LG_A_RDL);
        else if constexpr (hasD) LG_STEP_ASM("%[c0]", "%[c1]", LG_A_M0, LG_A_PIECE, "", "");
        else if constexpr (hasR) LG_STEP_ASM("%[c0]", "%[c1]", "", "", LG_A_RDH, LG_A_RDL);
        else LG_STEP_ASM("%[c0]", "%[c1]", "", "", "", "");
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // the pseudo step: no MFMAs; its read and its DMA piece issue as plain statements
      if constexpr (hasR) lg_read<RI, W2>(s);
      if constexpr (hasD) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(s.dma_l) : "memory");
        dma_lin<DOFF>(s.voff, s.dma_g);
      }
    }
    if constexpr (IT == LG_SYNC - 1) {
      // item NIT - 1 (the last item of this tile) has just been issued: aW now points into the next tile's stage
#pragma unroll
      for (int k = 0; k < 8; ++k) s.aW[k] += (unsigned)s.stage_delta;
      s.stage_delta = -s.stage_delta;
    }
    // ---- the per-tile barrier ...
    if constexpr (IT == LG_SYNC) {
      // the next tile's stage is complete (own DMA pieces landed, then everybody's), and every wave has ISSUED all its reads of
      // this tile (the last real one at step KS - 1 - PF): this tile's stage may be overwritten from here on
      unsigned long long tw = 0;
      if constexpr (TM) tw = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (TM) s.t_sync += __builtin_amdgcn_s_memtime() - tw;
      // (the next tile's item 0 is read only now, behind the barrier)
      if constexpr (kRd) lg_read<RI, W2>(s);
    }
    // ---- ... and a slice of the previous tile's epilogue
    if constexpr (kEp && lg_slice_step(IT)) lg_epilogue_slice<IT, OUT, ABL>(e, tile - 1);
    // ---- the pseudo step: the tile's sum goes to the transpose buffer (the previous tile's rows left it at steps 2, 3)
    if constexpr (kEp && IT == LG_KS) lg_epilogue_sum_write(e, s.accA, s.accB);
    __builtin_amdgcn_sched_barrier(0);
    lg_step<IT + 1, OUT, TM, ABL, NP>(s, e, tile);
  }
}
#undef LG_STEP_ASM
#undef LG_STEP_ASM2
#undef LG_STEP_ASM1
#undef LG_STEP_ASM1X
#undef LG_MFMA

// ---------------------------------------------------------------------------------------------------------------------------
// GEMM prologue (PRE = true): acc[t] (15 tiles of 32 output columns, AGPRs) += A[rows, K] · Wpre[:, K]^T in fp16 x 3, the rows'
// x = residual + bias + scale * acc then takes the place of the rows the plain kernel reads.  Same machinery as the tile loop —
// 2-stage ring of 64-KiB weight stages by linear LDS-DMA, one counted s_waitcnt vmcnt + s_barrier per stage, a fragment
// queue that runs on across stages, one asm statement per item — with the roles turned: a stage is a K-SLAB (32 k = two
// k16-steps of ALL 480 output columns, ldm_pack::pack_x3_slab_image), an item is (k16-step s, tile t) = 2 x 15 per stage, its three
// MFMAs go to the persistent accumulator of tile t, and the A operand — the wave's 32 rows, 8 halves per lane and k16-step — comes
// straight from global memory into registers, TWO stages ahead (plain loads into one of three register sets; the stage barrier's
// counted vmcnt leaves the youngest four of them in flight).
// Per workgroup: K / 32 stages x 64 KiB of weight slabs (linear2: 58 stages = 3.7 MB, the
// same bytes gemm16x3_k's 256 x 256 tile pulls for BOTH operands) and no launch, no fp32 round trip of the sum, no epilogue.
constexpr int LP_NT = 15, LP_NIT = 2 * LP_NT;   // items per stage
// a 3-deep fragment queue (the tile loop: 6): 24 registers — what it saves holds the third A register set, which hides the HBM latency of
// the A rows (22 - 27 us per launch, measurement variant 128); 3 items ahead are ~300 cycles of MFMA time, the LDS answers in ~130
constexpr int LP_PF = 3, LP_SYNC = LP_NIT - LP_PF;
static_assert(LP_NIT % LP_PF == 0, "queue slots must line up across stages");
// which of the 16 DMA pieces per wave and stage step IT carries: pieces 0 .. behind the barrier, the rest at the first steps of the next stage
constexpr int lp_piece(int IT) {
  if (IT > LP_SYNC) return IT - LP_SYNC - 1;
  if (IT + (LP_NIT - 1 - LP_SYNC) < 16) return IT + (LP_NIT - 1 - LP_SYNC);
  return -1;
}
static_assert(lp_piece(LP_SYNC) == -1 && lp_piece(LP_SYNC + 1) == 0 && lp_piece(0) == lp_piece(LP_NIT - 1) + 1, "16 pieces, in order, none at the barrier step");

struct LpState {
  f16x8 qh[LP_PF], ql[LP_PF];   // W hi / lo fragment queue
  unsigned aS[2];               // this lane's LDS address of k16-step s in the CURRENT stage (tile t: + t * 2 KiB)
  f16x8 fh[3][2], fl[3][2];     // A fragments [stage % 3][k16-step]: the current stage's, the next stage's (landed by the current
                                // stage's barrier) and the one after's (requested in this stage, in flight across its barrier: the
                                // rows come from HBM, a stage is ~1.5 us) — three register sets in rotation, no copies
  const __half *pa, *pal;       // this lane's row of A hi / lo + 8 * (lane / 32) halves
  size_t a_stage;               // halves from one 32-wide K stage of the A rows to the next: 32 (row-major rows) | one panel (panel-major)
  const char* img;              // slab image + wave * 16 KiB
  unsigned lds_w, voff;
  int stage_delta, n_stages, n_astages;   // stages of the image (a multiple of 3) / of them with real A columns
#ifdef LDM_LNGEMM_ABL_BUILD
  int a_hot;                    // (measurement build only)
#endif
  const char* dma_g;
  unsigned dma_l;
};

template <int IT, bool W2 = false>
__device__ __forceinline__ void lp_read(LpState& s) {   // item IT of the stage aS points at
  constexpr int sx = IT / LP_NT, t = IT % LP_NT;
  lg_dsr<t * 2048>(s.qh[IT % LP_PF], s.aS[sx]);
  if constexpr (!W2) lg_dsr<t * 2048 + LG_LO>(s.ql[IT % LP_PF], s.aS[sx]);
}
template <int I, bool W2 = false>
__device__ __forceinline__ void lp_prime(LpState& s) {
  if constexpr (I < LP_PF) {
    lp_read<I, W2>(s);
    lp_prime<I + 1, W2>(s);
  }
}
__device__ __forceinline__ void lp_dma_begin(LpState& s, int sd) {   // stage sd (clamped) -> ring slot sd & 1
  const int t = sd < s.n_stages ? sd : s.n_stages - 1;
  s.dma_g = s.img + (size_t)t * LG_STAGE;
  s.dma_l = s.lds_w + (unsigned)(sd & 1) * LG_STAGE;
}
// A fragments of stage sd (clamped) into register set P: k16-steps 2 sd, 2 sd + 1 -> 16 halves apart.  EXACTLY LP_A_LOADS vector
// memory instructions: the stage barrier's counted wait leaves that many outstanding.
constexpr int LP_A_LOADS = 4;   // (HI_ONLY — the one-product form, A is plain fp16: LP_A_LOADS / 2)
template <int P, bool HI_ONLY = false>
__device__ __forceinline__ void lp_load_a(LpState& s, int sd) {
#ifdef LDM_LNGEMM_ABL_BUILD   // measurement build, LDM_LNGEMM_ABL=128: every stage re-reads the A fragments of stage 0 (L2-resident)
  if (s.a_hot) sd = 0;
#endif
  const int t = sd < s.n_astages ? sd : s.n_astages - 1;   // (the zero slabs at the end of the image multiply the last real columns again)
  s.fh[P][0] = *reinterpret_cast<const f16x8*>(s.pa + (size_t)t * s.a_stage);
  s.fh[P][1] = *reinterpret_cast<const f16x8*>(s.pa + (size_t)t * s.a_stage + 16);
  if constexpr (!HI_ONLY) {
    s.fl[P][0] = *reinterpret_cast<const f16x8*>(s.pal + (size_t)t * s.a_stage);
    s.fl[P][1] = *reinterpret_cast<const f16x8*>(s.pal + (size_t)t * s.a_stage + 16);
  }
}

#define LP_MFMA(B, A) "v_mfma_f32_32x32x16_f16 %[c], " B ", " A ", %[c]\n\t"
#define LP_STEP_ASM(M0SET, PIECE, RD_HI, RD_LO, TAIL)                                                                             \
  asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" M0SET LP_MFMA("%[qh]", "%[xh]") LP_MFMA("%[qh]", "%[xl]") LP_MFMA("%[ql]", "%[xh]")    \
                   PIECE RD_HI RD_LO TAIL                                                                                          \
               : [c] "+a"(acc), [qh] "+v"(qh), [ql] "+v"(ql)                                                                       \
               : [xh] "v"(s.fh[SET][sx]), [xl] "v"(s.fl[SET][sx]), [aw] "v"(aw), [vo] "v"(s.voff), [dg] "s"(s.dma_g), [dl] "s"(s.dma_l),     \
                 [w] "n"(2 * (LP_PF - 1)), [ro] "n"(RO), [rl] "n"(RO + LG_LO), [doff] "n"(DOFF)                                    \
               : "memory")
#define LP_STEP_ASM2(M0SET, PIECE, RD_HI, TAIL)   /* W2: W_hi a_hi + W_hi a_lo */                                                 \
  asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" M0SET LP_MFMA("%[qh]", "%[xh]") LP_MFMA("%[qh]", "%[xl]") PIECE RD_HI TAIL            \
               : [c] "+a"(acc), [qh] "+v"(qh)                                                                                      \
               : [xh] "v"(s.fh[SET][sx]), [xl] "v"(s.fl[SET][sx]), [aw] "v"(aw), [vo] "v"(s.voff), [dg] "s"(s.dma_g), [dl] "s"(s.dma_l),     \
                 [w] "n"(LP_PF - 1), [ro] "n"(RO), [doff] "n"(DOFF)                                                                \
               : "memory")
#define LP_STEP_ASM1(M0SET, PIECE, RD_HI, TAIL)   /* one product: W_hi a_hi */                                                   \
  asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" M0SET LP_MFMA("%[qh]", "%[xh]") PIECE RD_HI TAIL                                    \
               : [c] "+a"(acc), [qh] "+v"(qh)                                                                                      \
               : [xh] "v"(s.fh[SET][sx]), [aw] "v"(aw), [vo] "v"(s.voff), [dg] "s"(s.dma_g), [dl] "s"(s.dma_l),                     \
                 [w] "n"(LP_PF - 1), [ro] "n"(RO), [doff] "n"(DOFF)                                                                \
               : "memory")

// the step at which the stage requests the A fragments of stage + 2: behind the last DMA piece of the slab the barrier certifies
constexpr int LP_A_STEP = 15;
static_assert(lp_piece(LP_A_STEP - 1) < 0 && lp_piece(LP_A_STEP) < 0 && LP_A_STEP < LP_SYNC, "A loads are the youngest vector memory operations at the barrier");

template <int IT, int SET, int NP = 3>   // SET = stage % 3: the A register set the stage multiplies; NP products per item (1: A hi only, plain fp16)
__device__ __forceinline__ void lp_step(LpState& s, f32x16* accs, int stage) {
  constexpr bool W2 = NP < 3;
  if constexpr (IT < LP_NIT) {
    constexpr int sx = IT / LP_NT, t = IT % LP_NT;
    constexpr int J = lp_piece(IT);
    constexpr bool hasD = J >= 0 && (!W2 || J < 8);   // (W2: the slab's hi half only)
    if constexpr (hasD && J == 0) lp_dma_begin(s, stage + 2);
    if constexpr (hasD && J > 0 && (J & 3) == 0) {
      s.dma_g += 4096;
      s.dma_l += 4096;
    }
    // the A fragments of stage + 2: requested behind this stage's last DMA piece, so that they are the LP_A_LOADS youngest vector
    // memory operations at the barrier, which does not wait for them (loads return in order: everything older has landed)
    if constexpr (IT == LP_A_STEP) lp_load_a<(SET + 2) % 3, NP == 1>(s, stage + 2);
    constexpr int RI = (IT + LP_PF) % LP_NIT;
    constexpr bool hasR = IT != LP_SYNC;
    constexpr int RO = (RI % LP_NT) * 2048;
    constexpr int DOFF = hasD ? (J & 3) * 1024 : 0;
    f32x16& acc = accs[t];
    f16x8& qh = s.qh[IT % LP_PF];
    f16x8& ql = s.ql[IT % LP_PF];
    const unsigned aw = s.aS[RI / LP_NT];
    __builtin_amdgcn_sched_barrier(0);
    // the stage's last item: 32 wait states behind its MFMAs, inside the statement (whatever hipcc places behind the stage loop —
    // its v_accvgpr_reads of the tiles — then finds every MFMA of the phase finished)
    static_assert(lp_piece(LP_NIT - 1) >= 0 && LP_NIT - 1 != LP_SYNC, "the last item carries a DMA piece and its reads");
    if constexpr (NP == 1) {
      if constexpr (IT == LP_NIT - 1) LP_STEP_ASM1(LG_A_M0, LG_A_PIECE, LG_A_RDH2 "\n\t", "s_nop 15\n\ts_nop 15");
      else if constexpr (hasD && hasR) LP_STEP_ASM1(LG_A_M0, LG_A_PIECE, LG_A_RDH2, "");
      else if constexpr (hasD) LP_STEP_ASM1(LG_A_M0, LG_A_PIECE, "", "");
      else if constexpr (hasR) LP_STEP_ASM1("", "", LG_A_RDH2, "");
      else LP_STEP_ASM1("", "", "", "");
    } else if constexpr (W2) {
      if constexpr (IT == LP_NIT - 1) LP_STEP_ASM2(LG_A_M0, LG_A_PIECE, LG_A_RDH2 "\n\t", "s_nop 15\n\ts_nop 15");
      else if constexpr (hasD && hasR) LP_STEP_ASM2(LG_A_M0, LG_A_PIECE, LG_A_RDH2, "");
      else if constexpr (hasD) LP_STEP_ASM2(LG_A_M0, LG_A_PIECE, "", "");
      else if constexpr (hasR) LP_STEP_ASM2("", "", LG_A_RDH2, "");
      else LP_STEP_ASM2("", "", "", "");
    } else if constexpr (IT == LP_NIT - 1) LP_STEP_ASM(LG_A_M0, LG_A_PIECE, LG_A_RDH, LG_A_RDL "\n\t", "s_nop 15\n\ts_nop 15");
    else if constexpr (hasD && hasR) LP_STEP_ASM(LG_A_M0, LG_A_PIECE, LG_A_RDH, LG_A_RDL, "");
    else if constexpr (hasD) LP_STEP_ASM(LG_A_M0, LG_A_PIECE, "", "", "");
    else if constexpr (hasR) LP_STEP_ASM("", "", LG_A_RDH, LG_A_RDL, "");
    else LP_STEP_ASM("", "", "", "", "");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (IT == LP_SYNC - 1) {   // the stage's last item has been requested: aS moves to the next stage's slot
      s.aS[0] += (unsigned)s.stage_delta;
      s.aS[1] += (unsigned)s.stage_delta;
      s.stage_delta = -s.stage_delta;
    }
    if constexpr (IT == LP_SYNC) {
      // the next stage's slab (own pieces) and its A fragments (requested a stage ago); the fragments of stage + 2 stay in flight
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP == 1 ? LP_A_LOADS / 2 : LP_A_LOADS) : "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (NP == 1) asm volatile("" : "+v"(s.fh[(SET + 1) % 3][0]), "+v"(s.fh[(SET + 1) % 3][1])::"memory");
      else asm volatile("" : "+v"(s.fh[(SET + 1) % 3][0]), "+v"(s.fh[(SET + 1) % 3][1]), "+v"(s.fl[(SET + 1) % 3][0]), "+v"(s.fl[(SET + 1) % 3][1])::"memory");   // (hipcc's own counted wait for them lands here)
      lp_read<RI, W2>(s);
    }
    __builtin_amdgcn_sched_barrier(0);
    lp_step<IT + 1, SET, NP>(s, accs, stage);
  }
}
#undef LP_STEP_ASM
#undef LP_STEP_ASM2
#undef LP_STEP_ASM1
#undef LP_MFMA

}  // namespace

// NPM / NPP: products per k16-step of the tile loop / of the GEMM prologue — 3 (split), 2 (weights fp16 only), 1 (plain fp16: the activation
// operand has no lo half either)
template <bool ADA, int OUT, bool TM = false, int ABL = 0, bool PRE = false, int NPM = 3, int NPP = 3>
__global__ __launch_bounds__(256, 1) void lngemm16x3_k(LnGemmArgs a) {
  constexpr bool W2 = NPM < 3, W2P = NPP < 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long t_k0 = 0, t_pro = 0, t_loop = 0;
  if constexpr (TM) t_k0 = __builtin_amdgcn_s_memtime();
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, hi = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(lds_char_ptr)smem;
  const unsigned voff = (unsigned)lane * 16;
  float* spar = reinterpret_cast<float*>(smem + LG_PAR_OFF);
  float* sbias = reinterpret_cast<float*>(smem + LG_BIAS_OFF);

  // ---- parameter tables -> LDS: multiplier | shift (zero beyond D: padded columns come out as exact zeros), bias
  for (int i = tid; i < 512; i += 256) {
    const bool in = i < a.D;
    spar[i] = in ? (ADA ? 1.0f + a.p0[i] : a.p0[i]) : 0.f;
    spar[512 + i] = in ? a.p1[i] : 0.f;
  }
  for (int i = tid; i < a.n_tiles * 32; i += 256) sbias[i] = (a.bias && i < a.N) ? a.bias[i] : 0.f;
  constexpr int NG = 58;
  const int row = blockIdx.x * 128 + wave * 32 + r;
  const int rrow = row < a.M ? row : a.M - 1;

  // ---- GEMM prologue: the rows are computed here instead of read (out_proj / linear2 in front of the LayerNorm that takes their sum)
  f32x16 pacc[PRE ? LP_NT : 1];
  if constexpr (PRE) {
    float* spb = reinterpret_cast<float*>(smem + LG_PBIAS_OFF);
    for (int i = tid; i < 512; i += 256) spb[i] = (a.pre_bias && i < a.D) ? a.pre_bias[i] : 0.f;
    LpState ps;
    // this wave's share of a 64-KiB stage: 16 KiB of hi | lo — or (W2) 8 KiB of the hi half, which is all the two-product form reads
    constexpr int WSH = W2P ? 8192 : 16384, WGR = W2P ? 2 : 4;
    ps.img = a.pre_img + wave * WSH;
    ps.lds_w = lds0 + wave * WSH;
    ps.voff = voff;
    ps.n_stages = a.pre_stages;
    ps.n_astages = a.pre_astages;
#ifdef LDM_LNGEMM_ABL_BUILD
    ps.a_hot = a.relu >> 8;   // (launch_lngemm16x3 passes the variant in the upper bits)
#endif
    ps.stage_delta = LG_STAGE;
    // A rows: row-major [M, pre_lda], or (pre_panel_stride != 0, r06) panel-major [K / 32 panels][rows][32 halves] as the ReLU epilogue
    // of linear1 writes them: a wave's 32 rows of a stage are 2 KiB contiguous instead of 32 x 64 B out of 32 different rows
    const size_t a_pitch = a.pre_panel_stride ? 32 : (size_t)a.pre_lda;
    ps.a_stage = a.pre_panel_stride ? a.pre_panel_stride / 2 : 32;
    ps.pa = a.preA + (size_t)rrow * a_pitch + hi * 8;
    ps.pal = a.preAlo + (size_t)rrow * a_pitch + hi * 8;
    for (int t = 0; t < 2; ++t)   // slabs 0 / 1 -> ring slots 0 / 1
#pragma unroll
      for (int k = 0; k < WGR; ++k) dma_lin4(voff, ps.img + (size_t)t * LG_STAGE + k * 4096, lds0 + t * LG_STAGE + wave * WSH + k * 4096);
    lp_load_a<0, NPP == 1>(ps, 0);
    lp_load_a<1, NPP == 1>(ps, 1);
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) ps.aS[sx] = lds0 + r * 64 + ((((sx << 1) | hi) ^ ((r >> 2) & 3)) << 4);
#pragma unroll
    for (int t = 0; t < LP_NT; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) pacc[t][k] = 0.f;
    lp_dma_begin(ps, 1);
    {  // (stage 0, steps 0 ..: the pieces of slab 1 once more, as in the tile loop)
      constexpr int J0 = lp_piece(0), pre = (J0 >> 2) - ((J0 & 3) == 0 ? 1 : 0);
      ps.dma_g += pre * 4096;
      ps.dma_l += pre * 4096;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (NPP == 1) asm volatile("" : "+v"(ps.fh[0][0]), "+v"(ps.fh[0][1]), "+v"(ps.fh[1][0]), "+v"(ps.fh[1][1])::"memory");
    else asm volatile("" : "+v"(ps.fh[0][0]), "+v"(ps.fh[0][1]), "+v"(ps.fl[0][0]), "+v"(ps.fl[0][1]), "+v"(ps.fh[1][0]), "+v"(ps.fh[1][1]),
                      "+v"(ps.fl[1][0]), "+v"(ps.fl[1][1])::"memory");
    lp_prime<0, W2P>(ps);
    for (int st = 0; st < a.pre_stages; st += 3) {   // three stage bodies, one per A register set (the ring slot follows aS);
      lp_step<0, 0, NPP>(ps, pacc, st);               // launch_lngemm16x3: a multiple of three stages (the image ends in zero slabs)
      lp_step<0, 1, NPP>(ps, pacc, st + 1);
      lp_step<0, 2, NPP>(ps, pacc, st + 2);
    }
    // the queue's trailing reads and the clamped re-load of the last slab are out, the last MFMAs have written their tiles, and every
    // wave is through with the ring: from here it belongs to the tile loop
    // (the accumulators were written by asm MFMAs hipcc cannot see: the wait states its v_accvgpr_reads need sit INSIDE the last
    //  item's statement of every stage — see LP_STEP_ASM's TAIL; r05's first run of this phase read the last tiles early, logits
    //  error 3e-2, and neither a statement owning all 240 AGPRs nor builtin MFMAs behind the loop kept hipcc's copies behind them)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- weights: tiles 0 / 1 -> stages 0 / 1 (this wave's 16 KiB of each).  Every workgroup visits the tiles in the SAME order.
  // (r05 negative result, profiles/r05_call3_4_*: a per-workgroup rotation of the order — 32 CUs of an XCD on 32 different
  //  tiles instead of all on the same one — is 4 % SLOWER, 950 vs 988 layouts/s: the lock-step stream is served by the L2 once
  //  per XCD, the rotated one is not.)
  constexpr int WSH_T = W2 ? 8192 : 16384, WGR_T = W2 ? 2 : 4;   // (as in the GEMM prologue: W2 fetches the hi half of a stage only)
  const char* img = a.img + wave * WSH_T;
  for (int t = 0; t < 2 && t < a.n_tiles; ++t)
#pragma unroll
    for (int k = 0; k < WGR_T; ++k) dma_lin4(voff, img + (size_t)t * LG_STAGE + k * 4096, lds0 + t * LG_STAGE + wave * WSH_T + k * 4096);

  // ---- the rows, raw, in accumulator layout: lane (row, hi) owns columns 8 g + 4 hi .. + 3 of every 8-column group g
  float4 v[NG];
  if constexpr (PRE) {   // x = residual + bias + scale * (A . Wpre^T): tile t of the prologue holds groups 4 t .. 4 t + 3
    const float* spb = reinterpret_cast<const float*>(smem + LG_PBIAS_OFF);
    const float* rs = a.pre_res + (size_t)rrow * a.D + hi * 4;
    float* po = (a.pre_out && row < a.M) ? a.pre_out + (size_t)row * a.D + hi * 4 : nullptr;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float4 pb = *reinterpret_cast<const float4*>(spb + g * 8 + hi * 4);
      const float4 rr = *reinterpret_cast<const float4*>(rs + g * 8);
      const f32x16& c = pacc[g >> 2];
      v[g].x = (c[(g & 3) * 4 + 0] * a.pre_scale + pb.x) + rr.x;
      v[g].y = (c[(g & 3) * 4 + 1] * a.pre_scale + pb.y) + rr.y;
      v[g].z = (c[(g & 3) * 4 + 2] * a.pre_scale + pb.z) + rr.z;
      v[g].w = (c[(g & 3) * 4 + 3] * a.pre_scale + pb.w) + rr.w;
      if (po) *reinterpret_cast<float4*>(po + g * 8) = v[g];
    }
  } else if (a.tokens) {  // x = emb[token] + pos[s]   (nn_lib.py:204,220)
    const float* e = a.emb + (size_t)a.tokens[rrow] * a.D + hi * 4;
    const float* p = a.pos + (size_t)(rrow % a.S) * a.D + hi * 4;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float4 x = *reinterpret_cast<const float4*>(e + g * 8), y = *reinterpret_cast<const float4*>(p + g * 8);
      v[g] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
  } else {
    const float* x = a.x + (size_t)rrow * a.ldx + hi * 4;
#pragma unroll
    for (int g = 0; g < NG; ++g) v[g] = *reinterpret_cast<const float4*>(x + g * 8);
  }
  // two-pass statistics over the row (this lane's half + lane ^ 32)
  float s1 = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) s1 += (v[g].x + v[g].y) + (v[g].z + v[g].w);
  s1 += __shfl_xor(s1, 32, 64);
  const float inv_d = 1.0f / (float)a.D;
  const float mean = s1 * inv_d;
  float s2 = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const float dx = v[g].x - mean, dy = v[g].y - mean, dz = v[g].z - mean, dw = v[g].w - mean;
    s2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  s2 += __shfl_xor(s2, 32, 64);
  const float rstd = 1.0f / sqrtf(s2 * inv_d + 1e-5f);
  __syncthreads();  // parameter tables visible
  // normalise, write the residual base (ADA), split into hi / lo fragments: groups 2 ks, 2 ks + 1 -> fragment ks.  The hi
  // fragment takes the place of the raw values it was made from (v[] dies pair by pair), the lo fragment goes to AGPRs.
  f16x8 xhi[LG_KS], xlo[LG_KS];
  float* yrow = (ADA && a.y32) ? a.y32 + (size_t)rrow * a.D + hi * 4 : nullptr;
#pragma unroll
  for (int ks = 0; ks < LG_KS; ++ks) {
    f16x8 fh, fl;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int g = 2 * ks + u;
      const float4 gm = *reinterpret_cast<const float4*>(spar + g * 8 + hi * 4);
      const float4 sh = *reinterpret_cast<const float4*>(spar + 512 + g * 8 + hi * 4);
      float4 y;
      y.x = (v[g].x - mean) * rstd * gm.x + sh.x;
      y.y = (v[g].y - mean) * rstd * gm.y + sh.y;
      y.z = (v[g].z - mean) * rstd * gm.z + sh.z;
      y.w = (v[g].w - mean) * rstd * gm.w + sh.w;
      if (yrow && row < a.M) *reinterpret_cast<float4*>(yrow + g * 8) = y;
      const float yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const _Float16 h = (_Float16)yy[c];
        fh[u * 4 + c] = h;
        if constexpr (NPM > 1) fl[u * 4 + c] = (_Float16)((yy[c] - (float)h) * kSplitLoScale);
      }
    }
    if (NPM == 1 && (ks & 1)) xhi[ks] = to_agpr4(fh);   // (plain fp16: no lo fragments; the odd steps' x fragments take their place in the AGPRs)
    else xhi[ks] = fh;
    if constexpr (NPM > 1) xlo[ks] = to_agpr4(fl);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- tile loop
  LgState s;
  s.xhi = xhi;
  s.xlo = xlo;
  s.img = img;
  s.lds_w = lds0 + wave * WSH_T;
  s.voff = voff;
  s.n_tiles = a.n_tiles;
  s.stage_delta = LG_STAGE;
#pragma unroll
  for (int k = 0; k < 8; ++k) s.aW[k] = lds0 + r * RKB + ((((k << 1) | hi) ^ (r & 15)) << 4);
  s.dma_g = img;   // (tile 0, steps 0 .. : the pieces of "tile 1" once more — what the prologue's DMA already brings)
  lg_dma_begin(s, 1);
  {  // the 4-KiB group of the first piece issued at step 0 (lg_step itself moves on at pieces 4, 8, 12)
    constexpr int J0 = lg_piece(0), pre = (J0 >> 2) - ((J0 & 3) == 0 ? 1 : 0);
    s.dma_g += pre * 4096;
    s.dma_l += pre * 4096;
  }
  LgEpi e;
  e.a_bias = lds0 + LG_BIAS_OFF + (lane & 7) * 16;
  {
    const unsigned tp = lds0 + LG_TP_OFF + wave * LG_TP_BYTES;
    e.a_tpw = tp + (r * LG_TP_LD + hi * 4) * 4;
    e.a_tpr = tp + ((lane >> 3) * LG_TP_LD + (lane & 7) * 4) * 4;
  }
  e.C0 = OUT == 0 ? reinterpret_cast<const char*>(a.C32) : reinterpret_cast<const char*>(a.C16);
  e.C1 = reinterpret_cast<const char*>(a.C16lo);
  e.b0 = e.C0; e.b1 = e.C1; e.colmask = 0;
  e.tstride = OUT == 0 ? 128 : (OUT == 1 && !a.panel_out) ? 64 : a.panel_stride;   // (OUT = 3: always panels)
  e.N = a.N; e.c4 = (lane & 7) * 4;
  e.out_scale = a.out_scale;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int orow = blockIdx.x * 128 + wave * 32 + p * 8 + (lane >> 3);
    e.rowmask[p] = __ballot(orow < a.M);
    // (launch_lngemm16x3 checks that M * ld * element size fits 32 bits)
    if (OUT == 2 || OUT == 3 || (OUT == 1 && a.panel_out))   // panel-major: a tile's 32 columns are ONE panel of 64-byte rows
      e.voff[p] = (unsigned)orow * 64u + (unsigned)e.c4 * 2u;
    else
      e.voff[p] = OUT == 0 ? ((unsigned)orow * (unsigned)a.ldc32 + (unsigned)e.c4) * 4u : ((unsigned)orow * (unsigned)a.ldc16 + (unsigned)e.c4) * 2u;
  }
  // every fragment back in its registers, hipcc's scoreboard drained (its own row loads / y32 stores), tiles 0 / 1 landed
#pragma unroll
  for (int k = 0; k < LG_KS; ++k) {
    if constexpr (NPM > 1) asm volatile("" : "+v"(xhi[k]), "+a"(xlo[k]));
    else if (k & 1) asm volatile("" : "+a"(xhi[k]));
    else asm volatile("" : "+v"(xhi[k]));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // (r05 calls 19 - 21, profiles/r05_call18_21_lngemm_store_cost.txt: of linear1's 187 us, 33 are its global stores — 6 with the same
  //  stores aimed at an L2-resident target, i.e. the cost is the 237 MB of hi / lo hidden rows on their way to HBM, not store issue;
  //  a phase offset between the lock-step workgroups changes nothing, non-temporal stores are 25 % slower.)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (TM) t_pro = __builtin_amdgcn_s_memtime();
  if constexpr (ABL != 0) {   // (timing variants: whatever they leave unwritten starts defined)
    for (int k = 0; k < 16; ++k) s.accA[k] = s.accB[k] = 0.f;
    for (int k = 0; k < LG_PF; ++k) s.qh[k] = s.ql[k] = xhi[k];
  }
  if constexpr (!(ABL & 2)) lg_prime<0, W2>(s);
  for (int t = 0; t < a.n_tiles; ++t) lg_step<0, OUT, TM, ABL, NPM>(s, e, t);
  if constexpr (TM) t_loop = __builtin_amdgcn_s_memtime();
  // the last tile's epilogue (its sum is in the transpose buffer): the same slices, each behind a full wait
  lg_epilogue_slice<1, OUT>(e, a.n_tiles - 1);
  lg_epilogue_slice<2, OUT>(e, a.n_tiles - 1);
  lg_epilogue_slice<3, OUT>(e, a.n_tiles - 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  lg_epilogue_slice<LG_SYNC + 1, OUT>(e, a.n_tiles - 1);
  lg_epilogue_slice<LG_SYNC + 2, OUT>(e, a.n_tiles - 1);
  lg_epilogue_slice<LG_SYNC + 3, OUT>(e, a.n_tiles - 1);
  lg_epilogue_slice<LG_SYNC + 4, OUT>(e, a.n_tiles - 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (TM) {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      atomicAdd(&g_lngemm_phase[0], 1ull);
      atomicAdd(&g_lngemm_phase[1], t_end - t_k0);
      atomicAdd(&g_lngemm_phase[2], t_pro - t_k0);
      atomicAdd(&g_lngemm_phase[3], s.t_sync);
      atomicAdd(&g_lngemm_phase[4], s.t_lgkm);
      atomicAdd(&g_lngemm_phase[5], t_end - t_loop);
    }
  }
}

void lngemm_phase_read(unsigned long long* out8) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lngemm_phase), 8 * sizeof(unsigned long long));
  unsigned long long z[8] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lngemm_phase), z, sizeof(z));
}

int launch_lngemm16x3(const LnGemmArgs& a, hipStream_t st) {
  // (everything LnGemmArgs documents or the kernel assumes is checked here, not only by ldm_create)
  if (a.D != 464 || a.n_tiles < 2 || (a.n_tiles & 1) || a.n_tiles * 32 > 2048 || a.N > a.n_tiles * 32 || (a.N & 3) || a.M < 1) return -1;
  if (a.tokens && a.S <= 0) return -1;
  // three output forms: fp32 (no ReLU) | ReLU + hi / lo fp16 rows | hi / lo fp16 panels without ReLU (q / k / v for kernels_attnout.hip);
  // the epilogue addresses with 32-bit byte offsets
  // products per k16-step (0 = 3): 3 split | 2 weights fp16 only | 1 plain fp16 (the tile loop: no x_lo fragments; the GEMM prologue: no preAlo)
  const int npm = a.np_main ? a.np_main : 3, npp = !a.pre_img ? npm : a.np_pre ? a.np_pre : 3;   // (no prologue: NPP follows NPM — one instantiation per form)
  const bool half_out = a.C16 != nullptr;
  const bool hi_only = half_out && !a.C16lo;                    // ReLU + plain fp16 panels (OUT = 3): linear1 in front of a one-product linear2
  const bool panel = half_out && a.panel_out && !a.relu;        // q / k / v (OUT = 2); panel_out with ReLU: linear1's hidden rows (OUT = 1)
  if (hi_only && (!a.relu || !a.panel_out || a.ada || a.C32)) return -1;
  if (half_out && a.panel_out && a.relu && ((a.N & 31) || (a.panel_stride & 15) || a.panel_stride < (size_t)a.M * 64 || (unsigned long long)a.M * 64 >= (1ull << 32)))
    return -1;
  if (panel ? (!a.C16lo || a.C32 || a.relu || !a.ada || (a.N & 31) || (a.panel_stride & 15) || a.panel_stride < (size_t)a.M * 64)
            : half_out ? ((!a.C16lo && !hi_only) || a.C32 || !a.relu || a.ada) : (!a.C32 || a.relu))
    return -1;
  if (panel ? ((unsigned long long)a.M * 64 >= (1ull << 32))
            : ((unsigned long long)a.M * (unsigned long long)(half_out ? a.ldc16 * 2 : a.ldc32 * 4) >= (1ull << 32)))
    return -1;
  const bool tm = knob_int("LDM_LNGEMM_TM", 0) != 0;   // (dev: the phase-timer instantiation, tools/lngemm_probe.py; read per launch: cheap, dev mode only)
  const bool pre = a.pre_img != nullptr;
  // GEMM prologue: a multiple of three 32-wide K slabs (zero slabs behind the pre_astages real ones), all d_model columns inside its 15 tiles, fp32 residual rows
  if (pre && (a.pre_stages < 3 || a.pre_stages % 3 || a.pre_astages < 2 || a.pre_astages > a.pre_stages || a.D > 32 * LP_NT || !a.preA ||
              (!a.preAlo && npp != 1) || !a.pre_res || a.tokens ||
              (a.pre_panel_stride ? ((a.pre_panel_stride & 15) || a.pre_panel_stride < (size_t)a.M * 64) : (a.pre_lda < 32 * a.pre_astages || (a.pre_lda & 7)))))
    return -1;
  // instantiations: <ADA, OUT, TM, ABL, PRE, NPM, NPP>; a launch without a GEMM prologue passes NPP = NPM (one instantiation per form)
  void (*kern)(LnGemmArgs) = nullptr;
#define LG_PICK(NPM_, NPP_)                                                                                                                           \
  kern = pre ? (panel ? lngemm16x3_k<true, 2, false, 0, true, NPM_, NPP_> : hi_only ? nullptr : half_out ? lngemm16x3_k<false, 1, false, 0, true, NPM_, NPP_> \
                : a.ada ? lngemm16x3_k<true, 0, false, 0, true, NPM_, NPP_> : lngemm16x3_k<false, 0, false, 0, true, NPM_, NPP_>)                         \
             : (panel ? lngemm16x3_k<true, 2, false, 0, false, NPM_, NPM_> : hi_only ? nullptr : half_out ? lngemm16x3_k<false, 1, false, 0, false, NPM_, NPM_> \
                : a.ada ? lngemm16x3_k<true, 0, false, 0, false, NPM_, NPM_> : lngemm16x3_k<false, 0, false, 0, false, NPM_, NPM_>)
  if (npm == 3 && npp == 3) LG_PICK(3, 3);
  else if (npm == 2 && npp == 2) LG_PICK(2, 2);
  // the hybrid mode's three forms beside mixed's in_proj: linear2 (plain fp16) in front of the two-product in_proj; linear1 in plain fp16 writing
  // plain-fp16 hidden panels; linear2 + head in plain fp16
  else if (npm == 2 && npp == 1 && pre && panel) kern = lngemm16x3_k<true, 2, false, 0, true, 2, 1>;
  else if (npm == 1 && !pre && hi_only) kern = lngemm16x3_k<false, 3, false, 0, false, 1, 1>;
  else if (npm == 1 && npp == 1 && pre && !half_out && !a.ada) kern = lngemm16x3_k<false, 0, false, 0, true, 1, 1>;
  else if (npm == 1 && !pre && !half_out && !a.ada) kern = lngemm16x3_k<false, 0, false, 0, false, 1, 1>;   // the head alone (behind the fused fp16 FFN)
#undef LG_PICK
  if (!kern) return -1;
  if (tm && !pre && !panel && !hi_only && npm == 3) kern = half_out ? lngemm16x3_k<false, 1, true> : a.ada ? lngemm16x3_k<true, 0, true> : lngemm16x3_k<false, 0, true>;
#ifdef LDM_LNGEMM_ABL_BUILD   // measurement build (tools/build_measurement_variants.py lngemm): compile-time timing variants of the loop
  static const int abl_knob = (int)knob_int("LDM_LNGEMM_ABL", 0);
  const int abl = (pre || panel || hi_only || npm != 3) ? 0 : abl_knob;
#define LG_ABL(n) case n: kern = half_out ? lngemm16x3_k<false, 1, false, n> : a.ada ? lngemm16x3_k<true, 0, false, n> : lngemm16x3_k<false, 0, false, n>; break;
  switch (abl) { LG_ABL(2) LG_ABL(4) LG_ABL(8) LG_ABL(6) LG_ABL(10) LG_ABL(12) LG_ABL(14) LG_ABL(16) LG_ABL(32) LG_ABL(64) default: break; }
#undef LG_ABL
#endif
  allow_big_lds((const void*)kern);
#ifdef LDM_LNGEMM_ABL_BUILD
  if (pre && abl_knob == 128) {
    LnGemmArgs b = a;
    b.relu |= 1 << 8;
    hipLaunchKernelGGL(kern, dim3((a.M + 127) / 128), dim3(256), LG_LDS, st, b);
    return 0;
  }
#endif
  hipLaunchKernelGGL(kern, dim3((a.M + 127) / 128), dim3(256), LG_LDS, st, a);
  return 0;
}

}  // namespace ldm
