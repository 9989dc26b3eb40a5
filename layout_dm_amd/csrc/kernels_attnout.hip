// Split (reference-precision) mode: self-attention AND its output projection in ONE launch, a layout's rows resident in the
// workgroup that owns it (r06):
//
//   Q[row, :] = P[row, :] + b_out + s_out * sum_h  softmax(q_h k_h^T / sqrt(dh)) v_h · Wo[:, h*dh .. ]^T
//
// (torch.nn.MultiheadAttention inside Block.forward, transformer_utils.py:175-178,197-204: the residual is taken on the
// NORMED x = P.)  It replaces attn16x3_k (kernels_attn16.hip) + the out_proj launch of gemm16x3_k: the per-head attention output
// never leaves the registers — the fast kernel's AttnCoreV -> SlabPair shape (ldm_pipes.h) with hi / lo split operands and three
// MFMAs per product — so att16 / att16lo (65 MB per 256 layouts, written and read back) do not exist, and the hi / lo split of q, k, v
// happens once, in in_proj's epilogue (kernels_lngemm.hip OUT = 2), not per (layout, head) on the way into LDS.
//
// One workgroup (4 waves) per layout; wave w owns query rows 32 w .. 32 w + 31 for everything:
//   S^T   = K_h · Q_w^T      D[key][query]   lane (query = lane & 31, g = lane >> 5): 64 scores of ITS query (the other 64 in lane ^ 32)
//   O^T   = V_h^T · P^T      D[d][query]     the P operand IS the lane's score registers (k-slot order = accumulator order)
//   out^T += Wo_h · O        D[col][query]   15 persistent 32-column tiles (240 accumulator registers), the O operand IS the lane's
//                                            O^T registers (Wo's K axis packed in k-slot order on the host)
// every product as hi·hi + lo·hi + hi·lo on v_mfma_f32_32x32x16_f16 (lo unscaled, ldm_kernels.h kSplitLoScale; P scaled by 2^10
// before its split as in attn16x3_k; Wo pre-scaled by a power of two per tensor, undone by out_scale).
//
// Operands.  in_proj writes q / k / v head-padded (58 -> 64) and PANEL-major: hi and lo arrays [96 panels][rows][16 halfs], panel
// (which * 8 + head) * 4 + d / 16 — a layout's 125 rows of a panel are 4 000 contiguous bytes.
//   K, V   whole-panel LDS-DMA (global_load_lds_dwordx4: no staging registers) into [panel][key][32 B] images; the DMA's per-lane
//          SOURCE address applies the bank swizzle (16-byte chunk ^ key bit 3; odd panels: key ^ 4) so that the image is linear in
//          LDS.  K fragments by ds_read_b128; V is needed TRANSPOSED (the contraction runs over keys): ds_read_b64_tr_b16 reads
//          a [4 keys][16 d] block per 16-lane group and hands each lane its d column — no transposed copy of V anywhere.
//   Q      straight from global memory into fragment registers (a wave's 32 rows x 32 B of a panel are 1 KiB contiguous).
//   Wo     k-step image (ldm_pack::pack_x3_kstep_image): per (head, k16-step) one 32-KiB stage = hi | lo of 480 rows x 32 B, by
//          linear LDS-DMA through a 3-slot ring; L2-resident (1 MiB per layer, every workgroup streams the same bytes).
// Schedule per head: S^T | softmax | P V | 4 out_proj stages, six workgroup barriers, each behind a COUNTED s_waitcnt vmcnt: the
// DMA of K / V of head h + 1 and of the ring stages runs 2 - 5 phases ahead of its consumer (issue order and counts below).
// LDS: K hi | lo 32 KiB, V hi | lo 32 KiB, ring 3 x 32 KiB = 160 KiB: one workgroup per CU; 240 accumulator + <= 256 arch registers.
// gfx950 only; geometry: d_model 464 (15 tiles), 8 heads of 58 (padded 64), S <= 128.
#include "ldm_dma.h"
#include "ldm_kernels.h"
#include "ldm_pipes.h"

namespace ldm {

namespace {

typedef __fp16 ao_h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 ao_f16x4 __attribute__((ext_vector_type(4)));

constexpr int AO_KH = 0, AO_VH = 32768, AO_RING = 65536, AO_SLOT = 32768, AO_LO = 16384;   // K lo / V lo at + AO_LO
constexpr int AO_LDS = AO_RING + 3 * AO_SLOT;   // 163 840
static_assert(AO_LDS <= 160 * 1024, "LDS budget");
constexpr int AO_NT = 15;                        // 32-column output tiles
constexpr int AO_RD = 2;                         // LDS read pipeline: fragments requested this many (3-MFMA) items ahead
constexpr int AO_UNIT = 8;                       // DMA pieces (1 KiB) per wave and unit (K, V, one ring stage) — the counted waits below

// one DMA unit = 8 pieces of this wave: two groups of four (one M0 write each)
__device__ __forceinline__ void ao_dma8(unsigned voff, const char* g0, unsigned l0, const char* g1, unsigned l1) {
  dma_lin4(voff, g0, l0);
  dma_lin4(voff, g1, l1);
}

template <int N>
__device__ __forceinline__ void ao_sync() {   // everything but the N youngest vector-memory operations of this wave landed; then everybody's
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int OFF>
__device__ __forceinline__ f16x8 ao_ldg128(unsigned voff, const char* sbase) {   // asm: hipcc must not count it (it would drain the DMA queue at the first use)
  f16x8 d;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
  return d;
}

__device__ __forceinline__ f16x8 ao_lds128(unsigned addr) {
  return *reinterpret_cast<const f16x8 __attribute__((address_space(3)))*>((lds_char_ptr)(size_t)addr);
}
__device__ __forceinline__ ao_f16x4 ao_tr(unsigned addr) {
  const ao_h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((ao_h16x4 __attribute__((address_space(3)))*)(lds_char_ptr)(size_t)addr);
  return __builtin_bit_cast(ao_f16x4, v);
}

// Every MFMA of the kernel is inline asm with its accumulator pinned to a register file: the scores / P V tiles in arch VGPRs ("v"),
// the 15 persistent out_proj tiles in AGPRs ("a").  (With builtin MFMAs hipcc chose the AGPR form for the score tiles and paid for it
// by moving out_proj tiles between the files: ~450 v_accvgpr_read / _write per head.)  Consequences of hiding them from hipcc:
//   * operands written by the VALU right in front (the fp16 casts of P) need wait states hipcc does not insert: NOPS_IN;
//   * a tile's last statement carries the wait states hipcc's VALU reads of the result need (8-pass MFMA): TAIL.
// a_lo·b_hi + a_hi·b_lo + a_hi·b_hi into ONE accumulator (attn16x3_k's order)
#define AO_MFMA "v_mfma_f32_32x32x16_f16 "
template <bool ZERO, bool NOPS_IN, bool TAIL>
__device__ __forceinline__ void ao_mfma3v(f32x16& c, const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl) {
  if constexpr (ZERO)
    asm volatile(AO_MFMA "%[c], %[al], %[bh], 0\n\t" AO_MFMA "%[c], %[ah], %[bl], %[c]\n\t" AO_MFMA "%[c], %[ah], %[bh], %[c]"
                 : [c] "=&v"(c) : [ah] "v"(ah), [al] "v"(al), [bh] "v"(bh), [bl] "v"(bl));
  else if constexpr (NOPS_IN && TAIL)
    asm volatile("s_nop 4\n\t" AO_MFMA "%[c], %[al], %[bh], %[c]\n\t" AO_MFMA "%[c], %[ah], %[bl], %[c]\n\t" AO_MFMA "%[c], %[ah], %[bh], %[c]\n\t"
                 "s_nop 15\n\ts_nop 3"
                 : [c] "+v"(c) : [ah] "v"(ah), [al] "v"(al), [bh] "v"(bh), [bl] "v"(bl));
  else if constexpr (NOPS_IN)
    asm volatile("s_nop 4\n\t" AO_MFMA "%[c], %[al], %[bh], %[c]\n\t" AO_MFMA "%[c], %[ah], %[bl], %[c]\n\t" AO_MFMA "%[c], %[ah], %[bh], %[c]"
                 : [c] "+v"(c) : [ah] "v"(ah), [al] "v"(al), [bh] "v"(bh), [bl] "v"(bl));
  else if constexpr (TAIL)
    asm volatile(AO_MFMA "%[c], %[al], %[bh], %[c]\n\t" AO_MFMA "%[c], %[ah], %[bl], %[c]\n\t" AO_MFMA "%[c], %[ah], %[bh], %[c]\n\t"
                 "s_nop 15\n\ts_nop 3"
                 : [c] "+v"(c) : [ah] "v"(ah), [al] "v"(al), [bh] "v"(bh), [bl] "v"(bl));
  else
    asm volatile(AO_MFMA "%[c], %[al], %[bh], %[c]\n\t" AO_MFMA "%[c], %[ah], %[bl], %[c]\n\t" AO_MFMA "%[c], %[ah], %[bh], %[c]"
                 : [c] "+v"(c) : [ah] "v"(ah), [al] "v"(al), [bh] "v"(bh), [bl] "v"(bl));
}

// hi = fp16(x), lo = fp16(x - float(hi)) for 8 values, as ONE asm statement (the instructions of kernels_lngemm.hip's epilogue):
// v_cvt_pk_f16_f32 for the hi pair, v_fma_mix_f32 (-hi * 1.0 + x: exact in fp32) and a second v_cvt_pk for the lo pair — 2 VALU
// instructions per value and, above all, ONE definition of hi.  Written in C++ ("h = (_Float16)x; l = (_Float16)(x - (float)h)" with
// x = o * inv) hipcc folded the cast of the product into v_fma_mixlo_f16 (one rounding from the exact product) for the hi that feeds
// l, but kept v_mul_f32 + v_cvt_pk_f16_f32 (two roundings) for the hi it hands to the MFMA: at near-ties the two differ by one fp16
// ulp, hi + lo is off by 2^-13 relative in one element of ~10^4, and the logits error of a pass was 5e-6 .. 8e-5 where the r05 kernels
// have 6e-7 (tests/test_attnout_gpu.py isolates it: one element of one row, off by exactly 2^-13 / 2^-14).  Being volatile asm the
// statement also stays where it is written: in FRONT of the barrier that separates it from the asm MFMAs reading its results
// (hipcc had sunk its own casts to one instruction in front of them — a VALU-write -> MFMA-read hazard it cannot see).
__device__ __forceinline__ void ao_split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
  typedef unsigned ao_u32x4 __attribute__((ext_vector_type(4)));
  ao_u32x4 h, l;
  float t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile(
      "v_cvt_pk_f16_f32 %[h0], %[x0], %[x1]\n\tv_cvt_pk_f16_f32 %[h1], %[x2], %[x3]\n\t"
      "v_cvt_pk_f16_f32 %[h2], %[x4], %[x5]\n\tv_cvt_pk_f16_f32 %[h3], %[x6], %[x7]\n\t"
      "v_fma_mix_f32 %[t0], -%[h0], 1.0, %[x0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %[t1], -%[h0], 1.0, %[x1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %[t2], -%[h1], 1.0, %[x2] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %[t3], -%[h1], 1.0, %[x3] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %[t4], -%[h2], 1.0, %[x4] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %[t5], -%[h2], 1.0, %[x5] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %[t6], -%[h3], 1.0, %[x6] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %[t7], -%[h3], 1.0, %[x7] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_cvt_pk_f16_f32 %[l0], %[t0], %[t1]\n\tv_cvt_pk_f16_f32 %[l1], %[t2], %[t3]\n\t"
      "v_cvt_pk_f16_f32 %[l2], %[t4], %[t5]\n\tv_cvt_pk_f16_f32 %[l3], %[t6], %[t7]"
      : [h0] "=&v"(h[0]), [h1] "=&v"(h[1]), [h2] "=&v"(h[2]), [h3] "=&v"(h[3]), [l0] "=&v"(l[0]), [l1] "=&v"(l[1]), [l2] "=&v"(l[2]),
        [l3] "=&v"(l[3]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6),
        [t7] "=&v"(t7)
      : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]));
  hi = __builtin_bit_cast(f16x8, h);
  lo = __builtin_bit_cast(f16x8, l);
}

}  // namespace

__global__ __launch_bounds__(256, 1) void attnout16x3_k(AttnOutArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, g = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(lds_char_ptr)smem;
  const int S = a.S;
  const size_t row0 = (size_t)blockIdx.x * S;
  const size_t PS = a.panel_stride;                       // bytes between panels
  const int q = wave * 32 + m;                            // this lane's query row inside the layout

  // ---- addresses
  // K / V whole-panel DMA: wave w moves panel w of the head (hi and lo: 4 + 4 pieces).  LDS slot (piece j, lane l) = row position
  // 32 j + l / 2, physical chunk l & 1; it receives key = position ^ (4 on odd panels), logical chunk = physical ^ key bit 3.
  const unsigned pos_l = (unsigned)lane >> 1;
  const unsigned voff_kv = ((pos_l ^ ((wave & 1) ? 4u : 0u)) << 5) | ((((unsigned)lane & 1u) ^ ((pos_l >> 3) & 1u)) << 4);
  const unsigned voff_lin = (unsigned)lane * 16;
  // per head: panels (which * 8 + head) * 4 + wave, rows row0 ..
  const char* gq_hi = a.qkv_hi + row0 * 32;               // + panel * PS
  const char* gq_lo = a.qkv_lo + row0 * 32;
  // fragment addresses (LDS bytes): K rows / Wo rows by (m, g); V blocks by 16-lane group
  const unsigned a_row = ((unsigned)m << 5) | ((((unsigned)g) ^ (((unsigned)m >> 3) & 1u)) << 4);          // even panels, Wo stages
  const unsigned a_row_odd = ((((unsigned)m) ^ 4u) << 5) | ((((unsigned)g) ^ (((unsigned)m >> 3) & 1u)) << 4);   // odd K panels
  const unsigned G = (unsigned)lane >> 4, sl = (unsigned)lane & 15u;
  const unsigned v_key0 = (4u * (G >> 1) + (sl >> 2)) ^ (4u * (G & 1u));
  const unsigned a_v1 = lds0 + (G & 1u) * 4096u + (v_key0 << 5) + (((sl & 3u) >> 1) << 4) + ((sl & 1u) << 3);
  const unsigned a_v2 = lds0 + (G & 1u) * 4096u + (v_key0 << 5) + ((((sl & 3u) >> 1) ^ 1u) << 4) + ((sl & 1u) << 3) + 256u;
  // Q fragments: this lane's row of panel ks, chunk g
  const unsigned voff_q = (unsigned)q * 32u + (unsigned)g * 16u;

  auto dma_kv = [&](int head, int which, unsigned lds_hi) {   // K (which = 1) / V (which = 2) of `head`: this wave's panel, hi then lo
    const size_t pn = (size_t)((which * 8 + head) * 4 + wave);
    ao_dma8(voff_kv, gq_hi + pn * PS, lds0 + lds_hi + wave * 4096, gq_lo + pn * PS, lds0 + lds_hi + AO_LO + wave * 4096);
  };
  auto dma_w = [&](int stage, int slot) {                      // ring stage (clamped: behind the last one a free slot is re-loaded)
    const int st = stage < 32 ? stage : 31;
    const char* gsrc = a.w_img + (size_t)st * AO_SLOT + wave * 8192;
    const unsigned l = lds0 + AO_RING + slot * AO_SLOT + wave * 8192;
    ao_dma8(voff_lin, gsrc, l, gsrc + 4096, l + 4096);
  };
  f16x8 qh[4], ql[4];
  auto load_q = [&](int head) {                                // 8 asm loads = one DMA unit's worth in the counted waits
    const size_t pn = (size_t)(head < 8 ? head : 7) * 4;
    const char* bh = gq_hi + pn * PS;
    const char* bl = gq_lo + pn * PS;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[ks] = ao_ldg128<0>(voff_q, bh + (size_t)ks * PS);
      ql[ks] = ao_ldg128<0>(voff_q, bl + (size_t)ks * PS);
    }
  };

  f32x16 pacc[AO_NT];
#pragma unroll
  for (int t = 0; t < AO_NT; ++t)
#pragma unroll
    for (int k = 0; k < 16; ++k) pacc[t][k] = 0.f;

  // ---- scores^T of one head (unscaled q · k): 4 key tiles x 4 k16-steps x 3 products; K panel ks of the image, odd panels row ^ 4
  f32x16 sc[4];
  // The asm MFMAs are scheduling barriers for hipcc, so the LDS reads are software-pipelined in the SOURCE: the fragments of item
  // i + AO_RD are requested in front of item i's MFMAs (first GPU run: read, s_waitcnt lgkmcnt(0), three MFMAs, read, ... — every LDS
  // round trip exposed, 38 k cycles per head where the MFMAs take 8.8 k).  hipcc still counts the waits (lgkmcnt(2 AO_RD - ..)).
  auto scores = [&]() {
    f16x8 kh[AO_RD + 1], kl[AO_RD + 1];
    auto rd = [&](int it) {   // item it = (kt, ks)
      const int kt = it >> 2, ks = it & 3;
      const unsigned ad = lds0 + AO_KH + ks * 4096 + kt * 1024 + ((ks & 1) ? a_row_odd : a_row);
      kh[it % (AO_RD + 1)] = ao_lds128(ad);
      kl[it % (AO_RD + 1)] = ao_lds128(ad + AO_LO);
    };
#pragma unroll
    for (int it = 0; it < AO_RD; ++it) rd(it);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      if (it + AO_RD < 16) rd(it + AO_RD);
      const int kt = it >> 2, ks = it & 3, b = it % (AO_RD + 1);
      if (ks == 0) ao_mfma3v<true, false, false>(sc[kt], kh[b], kl[b], qh[ks], ql[ks]);
      else if (it == 15) ao_mfma3v<false, false, true>(sc[kt], kh[b], kl[b], qh[ks], ql[ks]);
      else ao_mfma3v<false, false, false>(sc[kt], kh[b], kl[b], qh[ks], ql[ks]);
    }
  };

  // ---- prologue.  Vector-memory issue order (units of 8 per wave; the counted waits below rely on it):
  //   Q_0 K_0 | V_0 | (W_0,0 once more: keeps the counts of head 0 those of every head) | W_0,0 | W_0,1 | [Ba] W_0,2 | ...
  load_q(0);
  dma_kv(0, 1, AO_KH);
  dma_kv(0, 2, AO_VH);
  dma_w(0, 0);
  dma_w(0, 0);
  dma_w(1, 1);
  ao_sync<4 * AO_UNIT>();      // Ba(0): Q_0, K_0 landed (behind them: V_0, dummy, W_0,0, W_0,1)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qh[ks]), "+v"(ql[ks]));
  dma_w(2, 2);
  scores();

  int slot = 0;                // ring slot of stage 4 h (stage n lives in slot n % 3)
#pragma nounroll
  for (int h = 0; h < 8; ++h) {
    // ---- Bb(h): V_h landed (behind it: W_h-1,3  W_h,0  W_h,1  W_h,2); every wave is through with K_h
    ao_sync<4 * AO_UNIT>();
    load_q(h + 1);
    dma_kv(h < 7 ? h + 1 : 7, 1, AO_KH);
    // ---- softmax over the 128 keys of query m (64 here, 64 in lane ^ 32), fp32 — attn16x3_k's arithmetic
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (S < (kt + 1) * 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (key >= S) sc[kt][r] = -INFINITY;
        }
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = expf((sc[kt][r] - mx) * a.scale) * 1024.0f;
        sc[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    // ---- O^T = V^T · P^T: 2 d tiles x 8 k16-steps x 3 products (k-slot e of half g <-> accumulator register 8 hf + e)
    f32x16 o[2];
    {
      f16x8 vh8[AO_RD + 1], vl8[AO_RD + 1];
      auto rd = [&](int it) {   // item it = (kt, hf, dt)
        const int kt = it >> 2, hf = (it >> 1) & 1, dt = it & 1;
        const unsigned off = AO_VH + dt * 8192 + kt * 1024 + hf * 512;
        const ao_f16x4 h0 = ao_tr(a_v1 + off), h1 = ao_tr(a_v2 + off);
        const ao_f16x4 l0 = ao_tr(a_v1 + off + AO_LO), l1 = ao_tr(a_v2 + off + AO_LO);
        vh8[it % (AO_RD + 1)] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        vl8[it % (AO_RD + 1)] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
      };
#pragma unroll
      for (int it = 0; it < AO_RD; ++it) rd(it);
      f16x8 ph, pl;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        if (it + AO_RD < 16) rd(it + AO_RD);
        const int kt = it >> 2, hf = (it >> 1) & 1, dt = it & 1, b = it % (AO_RD + 1);
        if (dt == 0) {   // the P fragment of k16-step (kt, hf): registers 8 hf .. 8 hf + 7 of score tile kt
          const float x[8] = {sc[kt][hf * 8 + 0], sc[kt][hf * 8 + 1], sc[kt][hf * 8 + 2], sc[kt][hf * 8 + 3],
                              sc[kt][hf * 8 + 4], sc[kt][hf * 8 + 5], sc[kt][hf * 8 + 6], sc[kt][hf * 8 + 7]};
          ao_split8(x, ph, pl);
        }
        // (the casts of ph / pl sit right in front: wait states inside the statement; the last statement of a tile: its tail)
        if (it < 2) {
          asm volatile("s_nop 4" ::"v"(ph), "v"(pl));
          ao_mfma3v<true, false, false>(o[dt], vh8[b], vl8[b], ph, pl);
        } else if (it >= 14) ao_mfma3v<false, true, true>(o[dt], vh8[b], vl8[b], ph, pl);
        else ao_mfma3v<false, true, false>(o[dt], vh8[b], vl8[b], ph, pl);
      }
    }
    // the head's output as the B operand of its out_proj slabs: k16-step 2 dt + s <- registers 8 s .. 8 s + 7 of d tile dt
    f16x8 oh[4], ol[4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = o[dt][s2 * 8 + e] * inv;
        ao_split8(x, oh[2 * dt + s2], ol[2 * dt + s2]);
      }
    // ---- Bc(h): W_h,0 landed (behind it: W_h,1  W_h,2  Q_h+1 K_h+1); every wave is through with V_h
    ao_sync<4 * AO_UNIT>();
    dma_kv(h < 7 ? h + 1 : 7, 2, AO_VH);
    // ---- the head's four out_proj stages: out^T tile t += Wo[32 t .., k16-step] · O   (hi·hi + hi·lo + lo·hi)
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      if (st == 1) {          // Bd1: W_h,1 landed (behind it: W_h,2  Q K_h+1  V_h+1); stage 0's slot is free -> W_h,3
        ao_sync<4 * AO_UNIT>();
        dma_w(4 * h + 3, slot);
      } else if (st == 2) {   // Bd2: W_h,2 landed (behind it: Q K_h+1  V_h+1  W_h,3); stage 1's slot -> W_h+1,0
        ao_sync<4 * AO_UNIT>();
        dma_w(4 * h + 4, slot == 2 ? 0 : slot + 1);
      } else if (st == 3) {   // Bd3: W_h,3 landed (behind it: W_h+1,0); stage 2's slot -> W_h+1,1
        ao_sync<1 * AO_UNIT>();
        dma_w(4 * h + 5, slot == 0 ? 2 : slot - 1);
      }
      const int sl_st = st == 0 ? slot : st == 1 ? (slot == 2 ? 0 : slot + 1) : st == 2 ? (slot == 0 ? 2 : slot - 1) : slot;
      const unsigned aw = lds0 + AO_RING + (unsigned)sl_st * AO_SLOT + a_row;
      f16x8 wh[AO_RD + 1], wl[AO_RD + 1];
#pragma unroll
      for (int t = 0; t < AO_RD; ++t) {
        wh[t] = ao_lds128(aw + t * 1024);
        wl[t] = ao_lds128(aw + t * 1024 + AO_LO);
      }
#pragma unroll
      for (int t = 0; t < AO_NT; ++t) {
        if (t + AO_RD < AO_NT) {
          wh[(t + AO_RD) % (AO_RD + 1)] = ao_lds128(aw + (t + AO_RD) * 1024);
          wl[(t + AO_RD) % (AO_RD + 1)] = ao_lds128(aw + (t + AO_RD) * 1024 + AO_LO);
        }
        // (asm with an AGPR-pinned accumulator: with builtin MFMAs hipcc moved half of the 15 tiles between the register files
        //  in every head, 450 v_accvgpr_read / _write per iteration)
        asm volatile(AO_MFMA "%[c], %[wh], %[oh], %[c]\n\t" AO_MFMA "%[c], %[wh], %[ol], %[c]\n\t" AO_MFMA "%[c], %[wl], %[oh], %[c]"
                     : [c] "+a"(pacc[t])
                     : [wh] "v"(wh[t % (AO_RD + 1)]), [wl] "v"(wl[t % (AO_RD + 1)]), [oh] "v"(oh[st]), [ol] "v"(ol[st]));
      }
    }
    slot = slot == 2 ? 0 : slot + 1;   // stage 4 (h + 1) = 4 h + 4 -> slot + 4 mod 3
    // ---- Ba(h + 1): Q_h+1, K_h+1 landed (behind them: V_h+1  W_h,3  W_h+1,0  W_h+1,1); stage 3's slot -> W_h+1,2
    ao_sync<4 * AO_UNIT>();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qh[ks]), "+v"(ql[ks]));
    dma_w(4 * h + 6, slot == 0 ? 2 : slot - 1);
    if (h < 7) scores();
  }
  // (the clamped re-loads behind the last head: nothing may land after the workgroup is gone; and the asm MFMAs of the last stage,
  //  which hipcc cannot see, have written their tiles before its v_accvgpr_reads below)
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  // ---- epilogue: Q = P + b_out + out_scale * acc, in accumulator layout (lane (query, g) owns columns 8 gq + 4 g .. + 3 of each group)
  if (q < S) {
    const float* prow = a.res + (row0 + q) * a.D + g * 4;
    float* orow = a.out + (row0 + q) * a.D + g * 4;
    const float* bias = a.bias + g * 4;
#pragma unroll
    for (int t = 0; t < AO_NT; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int col = t * 32 + gq * 8;
        if (col + 8 <= a.D) {
          const float4 r = *reinterpret_cast<const float4*>(prow + col);
          const float4 b = *reinterpret_cast<const float4*>(bias + col);
          float4 y;
          y.x = (pacc[t][gq * 4 + 0] * a.out_scale + b.x) + r.x;
          y.y = (pacc[t][gq * 4 + 1] * a.out_scale + b.y) + r.y;
          y.z = (pacc[t][gq * 4 + 2] * a.out_scale + b.z) + r.z;
          y.w = (pacc[t][gq * 4 + 3] * a.out_scale + b.w) + r.w;
          *reinterpret_cast<float4*>(orow + col) = y;
        }
      }
  }
}

bool attnout16x3_supported(int S, int H, int dh, int D) { return S >= 1 && S <= 128 && H == 8 && dh == 58 && D == 464; }

int launch_attnout16x3(const AttnOutArgs& a, int B, hipStream_t st) {
  if (!attnout16x3_supported(a.S, 8, 58, a.D) || B < 1 || (a.panel_stride & 15) || !a.qkv_hi || !a.qkv_lo || !a.w_img) return -1;
  allow_big_lds((const void*)attnout16x3_k);
  hipLaunchKernelGGL(attnout16x3_k, dim3(B), dim3(256), AO_LDS, st, a);
  return 0;
}

}  // namespace ldm
