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
// Operands.  in_proj writes q / k / v head-padded (58 -> 64) and PANEL-major: hi and lo arrays [48 panels][rows][32 halfs], panel
// (which * 8 + head) * 2 + d / 32 — a layout's 125 rows of a panel are 8 000 contiguous bytes, 16 rows are one 1-KiB DMA piece.
//   K, V   whole-head LDS-DMA (global_load_lds_dwordx4: no staging registers); the DMA's per-lane SOURCE address permutes the 16-byte
//          chunks inside each 1-KiB piece, the destination is linear:
//            K image  [d / 32][key][64 B], chunk c of a key at c ^ ((key >> 2) & 3)  (the W2-slab format of the fused FFN: conflict-free
//                     ds_read_b128 by (key, k half) lanes);
//            V image  [d / 32][key / 4][(d / 16) & 1][key % 4][32 B]: V is needed TRANSPOSED (the contraction runs over keys) and
//                     ds_read_b64_tr_b16 does that in the LDS pipe — a 16-lane group reads a [4 keys][16 d] block and every lane
//                     receives its d column (cdna_hip_programming.md T10).  In this image the four blocks of one read — (d half 0 / 1)
//                     x (key quad q, q + 1): exactly what the MFMA A operand's lanes 0-15 / 16-31 / 32-47 / 48-63 hold — are 512
//                     CONTIGUOUS bytes, lane l at + 8 l: the guide's conflict-free form.  (r06 first form: [d / 16][key][32 B] with
//                     an XOR swizzle, conflict-free under the plain 64-bank model of MI355X_MICROARCH.md, ran the P V phase at
//                     7 400 cycles per head instead of ~1 700: the transpose read has its own conflict classes.)
//   Q      straight from global memory into fragment registers.
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
constexpr int AO_RDP = 1;                        // ... in the P V phase: this many (6-MFMA) key steps ahead
constexpr int AO_UNIT = 8;                       // DMA pieces (1 KiB) per wave and unit (K, V, one ring stage) — the counted waits below

// one DMA unit = 8 pieces of this wave: two groups of four (one M0 write each)
__device__ __forceinline__ void ao_dma8(unsigned voff, const char* g0, unsigned l0, const char* g1, unsigned l1) {
  dma_lin4(voff, g0, l0);
  dma_lin4(voff, g1, l1);
}

__device__ unsigned long long g_attnout_phase[24];   // TM instantiation (tools/attnout_probe.py phases): see attnout_phase_read
template <int N, bool TM = false>
__device__ __forceinline__ void ao_sync(unsigned long long* tw = nullptr, unsigned long long* tb = nullptr) {   // everything but the N youngest vector-memory operations of this wave landed; then everybody's
  unsigned long long t0 = 0, t1 = 0;
  if constexpr (TM) t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
  if constexpr (TM) t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (TM) {
    *tw += t1 - t0;
    *tb += __builtin_amdgcn_s_memtime() - t1;
  }
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

// One (kt, hf) key step of P V for BOTH d tiles as ONE statement: six MFMAs (o0 += V0 P, o1 += V1 P, each as lo·hi + hi·lo + hi·hi) with the
// hi / lo split of the NEXT key step's probabilities (ao_split8's 16 VALU instructions) placed BETWEEN them.  A wave issues in order: behind three
// dependent MFMAs nothing else gets out for ~96 cycles, then the VALU block runs with the matrix pipe idle — the first form of this phase (split,
// then two 3-MFMA statements) took 3 900 cycles per head where its 48 MFMAs take 1 540; between the MFMAs the VALU issues in their shadows.
// The operands the MFMAs read (ph, pl, the V fragments) were written at least one whole statement earlier; what the VALU writes here (hn, ln) is read
// by the NEXT statement.  FIRST: the accumulators start from zero.  SPLIT = false (the last key step): no next split.  TAIL: the wait states hipcc's
// VALU reads of o0 / o1 need behind MFMAs it cannot see.
template <bool FIRST, bool SPLIT, bool TAIL>
__device__ __forceinline__ void ao_pv_pair(f32x16& o0, f32x16& o1, const f16x8& vh0, const f16x8& vl0, const f16x8& vh1, const f16x8& vl1,
                                           const f16x8& ph, const f16x8& pl, const float (&x)[8], f16x8& hn, f16x8& ln) {
  typedef unsigned ao_u32x4 __attribute__((ext_vector_type(4)));
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  float t0, t1, t2, t3, t4, t5, t6, t7;
#define AO_V1 "v_cvt_pk_f16_f32 %[h0], %[x0], %[x1]\n\tv_cvt_pk_f16_f32 %[h1], %[x2], %[x3]\n\tv_cvt_pk_f16_f32 %[h2], %[x4], %[x5]\n\tv_cvt_pk_f16_f32 %[h3], %[x6], %[x7]\n\t"
#define AO_V2 "v_fma_mix_f32 %[t0], -%[h0], 1.0, %[x0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %[t1], -%[h0], 1.0, %[x1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
              "v_fma_mix_f32 %[t2], -%[h1], 1.0, %[x2] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %[t3], -%[h1], 1.0, %[x3] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
#define AO_V3 "v_fma_mix_f32 %[t4], -%[h2], 1.0, %[x4] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %[t5], -%[h2], 1.0, %[x5] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
              "v_fma_mix_f32 %[t6], -%[h3], 1.0, %[x6] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %[t7], -%[h3], 1.0, %[x7] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
#define AO_V4 "v_cvt_pk_f16_f32 %[l0], %[t0], %[t1]\n\tv_cvt_pk_f16_f32 %[l1], %[t2], %[t3]\n\tv_cvt_pk_f16_f32 %[l2], %[t4], %[t5]\n\tv_cvt_pk_f16_f32 %[l3], %[t6], %[t7]\n\t"
#define AO_PV(C0A, C0B, V1, V2, V3, V4, TAILS)                                                                                                  \
  asm volatile(AO_MFMA "%[o0], %[vl0], %[ph], " C0A "\n\t" V1 AO_MFMA "%[o0], %[vh0], %[pl], %[o0]\n\t" V2 AO_MFMA "%[o0], %[vh0], %[ph], %[o0]\n\t" V3 \
                   AO_MFMA "%[o1], %[vl1], %[ph], " C0B "\n\t" V4 AO_MFMA "%[o1], %[vh1], %[pl], %[o1]\n\t" AO_MFMA "%[o1], %[vh1], %[ph], %[o1]" TAILS \
               : [o0] "+v"(o0), [o1] "+v"(o1), [h0] "=&v"(h0), [h1] "=&v"(h1), [h2] "=&v"(h2), [h3] "=&v"(h3), [l0] "=&v"(l0), [l1] "=&v"(l1),          \
                 [l2] "=&v"(l2), [l3] "=&v"(l3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),        \
                 [t6] "=&v"(t6), [t7] "=&v"(t7)                                                                                                      \
               : [vh0] "v"(vh0), [vl0] "v"(vl0), [vh1] "v"(vh1), [vl1] "v"(vl1), [ph] "v"(ph), [pl] "v"(pl), [x0] "v"(x[0]), [x1] "v"(x[1]),         \
                 [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]))
  if constexpr (FIRST) {
    static_assert(SPLIT && !TAIL, "the first key step splits the second one's probabilities");
    AO_PV("0", "0", AO_V1, AO_V2, AO_V3, AO_V4, "");
  } else if constexpr (SPLIT) {
    AO_PV("%[o0]", "%[o1]", AO_V1, AO_V2, AO_V3, AO_V4, "");
  } else {
    static_assert(TAIL, "the last key step carries the tail");
    AO_PV("%[o0]", "%[o1]", "", "", "", "", "\n\ts_nop 15\n\ts_nop 3");
    h0 = h1 = h2 = h3 = l0 = l1 = l2 = l3 = 0;
  }
#undef AO_PV
#undef AO_V1
#undef AO_V2
#undef AO_V3
#undef AO_V4
  const ao_u32x4 hh = {h0, h1, h2, h3}, ll = {l0, l1, l2, l3};
  hn = __builtin_bit_cast(f16x8, hh);
  ln = __builtin_bit_cast(f16x8, ll);
}

}  // namespace

// W2: two-product out_proj (Wo fp16 only: its lo half is neither read nor multiplied).  FFN (the hybrid mode): the block's plain-fp16 FFN runs on the
// same accumulator tiles behind the attention — residual add in registers, LayerNorm-2, ldm_pipes.h FfnStream, ONE store of the block's output rows —
// instead of the epilogue below and a launch of kernels_ffn16.hip: the attention block's output Q never reaches memory
template <bool TM, bool W2 = false, bool FFN = false>
__global__ __launch_bounds__(256, 1) void attnout16x3_k(AttnOutArgs a) {
  // (TM: cycles in the counted vmcnt waits [0..5] and in the barriers behind them [6..11] per sync site Ba Bb Bc Bd1 Bd2 Bd3, the
  //  head loop [12], the epilogue [13], the whole kernel [14], workgroups [15])
  unsigned long long tw[6] = {0, 0, 0, 0, 0, 0}, tb[6] = {0, 0, 0, 0, 0, 0}, t_k0 = 0, t_l0 = 0, t_l1 = 0;
  unsigned long long tp[5] = {0, 0, 0, 0, 0}, tq = 0;   // [16..20]: softmax | P V | O split | out_proj stages (without their syncs) | scores
#define AO_T0() do { if constexpr (TM) tq = __builtin_amdgcn_s_memtime(); } while (0)
#define AO_T1(i) do { if constexpr (TM) tp[i] += __builtin_amdgcn_s_memtime() - tq; } while (0)
  if constexpr (TM) t_k0 = __builtin_amdgcn_s_memtime();
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
  // K / V whole-head DMA, 16 + 16 pieces of 1 KiB (16 keys x 64 B of one panel) per operand: wave w moves pieces 4 w .. 4 w + 3 of the
  // hi and of the lo image = panel w >> 1, keys 64 (w & 1) + 16 j.  LDS slot l of a piece receives
  //   K: key l >> 2, logical chunk (l & 3) ^ ((key >> 2) & 3)        V: key 4 (l >> 4) + ((l >> 1) & 3), chunk 2 ((l >> 3) & 1) + (l & 1)
  const unsigned lu = (unsigned)lane;
  const unsigned voff_k = ((lu >> 2) << 6) | (((lu & 3u) ^ ((lu >> 4) & 3u)) << 4);
  const unsigned voff_v = ((4u * (lu >> 4) + ((lu >> 1) & 3u)) << 6) | ((2u * ((lu >> 3) & 1u) + (lu & 1u)) << 4);
  const unsigned voff_lin = lu * 16;
  // per head: panels (which * 8 + head) * 2 + (wave >> 1), rows row0 + 64 (wave & 1) ..
  const char* gq_hi = a.qkv_hi + row0 * 64;               // + panel * PS
  const char* gq_lo = a.qkv_lo + row0 * 64;
  // fragment addresses (LDS bytes).  K rows by (m, g): + (ks >> 1) * 8 KiB + kt * 2 KiB, chunk (2 (ks & 1) + g) ^ ((m >> 2) & 3);
  // Wo stage rows by (m, g): 32-byte rows, chunk g ^ ((m >> 3) & 1)
  const unsigned a_k0 = ((unsigned)m << 6) | ((((unsigned)g) ^ (((unsigned)m >> 2) & 3u)) << 4);           // even k16-steps
  const unsigned a_k1 = ((unsigned)m << 6) | (((2u + (unsigned)g) ^ (((unsigned)m >> 2) & 3u)) << 4);      // odd k16-steps
  const unsigned a_row = ((unsigned)m << 5) | ((((unsigned)g) ^ (((unsigned)m >> 3) & 1u)) << 4);
  // V transpose reads: lane l at + 8 l of the 512-byte run of (d tile, 8 keys)
  const unsigned a_v = lds0 + lu * 8;
  // Q fragments: this lane's row of panel ks >> 1, chunk 2 (ks & 1) + g
  const unsigned voff_q = (unsigned)q * 64u + (unsigned)g * 16u;

  auto dma_kv = [&](int head, int which, unsigned lds_hi) {   // K (which = 1) / V (which = 2) of `head`: this wave's 4 + 4 pieces, hi then lo
    const size_t pn = (size_t)((which * 8 + head) * 2 + (wave >> 1));
    const size_t go = pn * PS + (size_t)(wave & 1) * 4096;
    ao_dma8(which == 1 ? voff_k : voff_v, gq_hi + go, lds0 + lds_hi + wave * 4096, gq_lo + go, lds0 + lds_hi + AO_LO + wave * 4096);
  };
  auto dma_w = [&](int stage, int slot) {                      // ring stage (clamped: behind the last one a free slot is re-loaded)
    const int st = stage < 32 ? stage : 31;
    const char* gsrc = a.w_img + (size_t)st * AO_SLOT + wave * 8192;
    const unsigned l = lds0 + AO_RING + slot * AO_SLOT + wave * 8192;
    ao_dma8(voff_lin, gsrc, l, gsrc + 4096, l + 4096);
  };
  // One residual-row unit of the LAST head (8 rows of this wave = 8 pieces: what the counted waits expect of a unit).  The last head's K_8 / V_8 /
  // W_8,* units have nothing left to fetch; instead of re-loading (r06 first form) they bring the epilogue's residual rows into the buffers the head
  // frees — K image, V image, the ring slots — so that the HBM reads of P overlap the last head's MFMAs instead of joining the epilogue's all-CU burst
  // (119 MB in 23 us = 5.1 TB/s).  Row r of a column half: 1 KiB at lds_base + (r & 7) KiB, 16-byte chunk c at c ^ (r & 15).
  constexpr int ND = 464;                  // launch_attnout16x3 checks D == 464
  const int nrow = S - wave * 32 < 32 ? S - wave * 32 : 32;   // rows of this wave that exist (wave-uniform)
  const char* pres = reinterpret_cast<const char*>(a.res + (row0 + (size_t)(wave * 32)) * ND);   // this wave's residual rows
  auto dma_prow = [&](int half, int r0, unsigned lds_base) {
    // (the row arithmetic hangs on an opaque copy of r0: hipcc otherwise hoists the 40 row addresses of the five units out of the head loop
    //  and keeps them alive across a body that already fills the register file — 400 SGPR spills, the head loop 8 % slower)
    int r0v = r0;
    asm volatile("" : "+s"(r0v));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // rows that do not exist: a valid row again, never stored.  A wave WITHOUT rows (S <= 32 wave) has nrow <= 0: rr is negative and
      // names the layout's last real row from this wave's base — in SIGNED arithmetic (r06 fix: the unsigned product sent S = 50's
      // waves 2 / 3 four gigabytes up; tests/test_hip_parity.py short_sequence passed or faulted with the process's memory map)
      const int r = r0v + j, rr = r < nrow ? r : nrow - 1;
      unsigned lc = lu ^ (unsigned)((r0 + j) & 15);                              // logical chunk that lands in physical chunk `lane`
      if (half == 1) lc = lc < 52u ? lc : 51u;                                   // (chunks beyond the row: a valid chunk again, never read)
      const char* src = pres + (long)rr * (long)(ND * 4) + half * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lc * 16u), "s"(src), "s"(lds_base + (unsigned)j * 1024u) : "memory");
    }
  };
  // where the epilogue finds them (absolute LDS byte addresses of this wave's 8-row groups; ring slots of head 7: (4 * 7) % 3 = 1 -> stage 1 in slot
  // 2, stage 2 in slot 0, stage 3 in slot 1: tests/test_attnout_layout.py replays the slot arithmetic)
  const unsigned grpA[4] = {lds0 + AO_KH + wave * 8192u, lds0 + AO_VH + wave * 8192u, lds0 + AO_RING + 2u * AO_SLOT + wave * 8192u,
                            lds0 + AO_RING + 0u * AO_SLOT + wave * 8192u};
  const unsigned grpB0 = lds0 + AO_RING + 1u * AO_SLOT + wave * 8192u;

  f16x8 qh[4], ql[4];
  auto load_q = [&](int head) {                                // 8 asm loads = one DMA unit's worth in the counted waits
    const size_t pn = (size_t)(head < 8 ? head : 7) * 2;
    const char* bh = gq_hi + pn * PS;
    const char* bl = gq_lo + pn * PS;
    qh[0] = ao_ldg128<0>(voff_q, bh);
    ql[0] = ao_ldg128<0>(voff_q, bl);
    qh[1] = ao_ldg128<32>(voff_q, bh);
    ql[1] = ao_ldg128<32>(voff_q, bl);
    qh[2] = ao_ldg128<0>(voff_q, bh + PS);
    ql[2] = ao_ldg128<0>(voff_q, bl + PS);
    qh[3] = ao_ldg128<32>(voff_q, bh + PS);
    ql[3] = ao_ldg128<32>(voff_q, bl + PS);
  };

  f32x16 pacc[AO_NT];
#pragma unroll
  for (int t = 0; t < AO_NT; ++t)
#pragma unroll
    for (int k = 0; k < 16; ++k) pacc[t][k] = 0.f;

  // ---- scores^T of one head (unscaled q · k): 4 key tiles x 4 k16-steps x 3 products
  f32x16 sc[4];
  // The asm MFMAs are scheduling barriers for hipcc, so the LDS reads are software-pipelined in the SOURCE: the fragments of item
  // i + AO_RD are requested in front of item i's MFMAs (first GPU run: read, s_waitcnt lgkmcnt(0), three MFMAs, read, ... — every LDS
  // round trip exposed, 38 k cycles per head where the MFMAs take 8.8 k).  hipcc still counts the waits (lgkmcnt(2 AO_RD - ..)).
  auto scores = [&]() {
    f16x8 kh[AO_RD + 1], kl[AO_RD + 1];
    auto rd = [&](int it) {   // item it = (kt, ks)
      const int kt = it >> 2, ks = it & 3;
      const unsigned ad = lds0 + AO_KH + (ks >> 1) * 8192 + kt * 2048 + ((ks & 1) ? a_k1 : a_k0);
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
  ao_sync<4 * AO_UNIT, TM>(&tw[0], &tb[0]);      // Ba(0): Q_0, K_0 landed (behind them: V_0, dummy, W_0,0, W_0,1)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qh[ks]), "+v"(ql[ks]));
  dma_w(2, 2);
  scores();

  int slot = 0;                // ring slot of stage 4 h (stage n lives in slot n % 3)
  if constexpr (TM) t_l0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int h = 0; h < 8; ++h) {
    // ---- Bb(h): V_h landed (behind it: W_h-1,3  W_h,0  W_h,1  W_h,2); every wave is through with K_h
    ao_sync<4 * AO_UNIT, TM>(&tw[1], &tb[1]);
    load_q(h + 1);
    if (h < 7) dma_kv(h + 1, 1, AO_KH);
    else dma_prow(0, 0, grpA[0]);
    AO_T0();
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
    // p = 2^10 exp((s - mx) scale) as ONE v_fma + ONE v_exp_f32 per score: 2^(s c + (10 - mx c)), c = scale log2(e).  (expf — OCML's
    // range-reduced form, ~10 instructions per call — made this phase 1 350 VALU instructions per head and wave; hipcc spread them
    // over the P V items, 7 400 cycles where the 48 MFMAs take 1 540.)  The rounding of 10 - mx c is common to the row's 128 keys and
    // cancels in p / sum; the argument's own rounding is 2^-24 |arg|: 6e-8 for the keys that carry the row, < 2e-6 relative on
    // probabilities below 2^-30.
    const float cexp = a.scale * 1.4426950408889634f;
    const float nb = 10.0f - mx * cexp;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][r], cexp, nb));
        sc[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    AO_T1(0);
    AO_T0();
    // ---- O^T = V^T · P^T: 2 d tiles x 8 k16-steps x 3 products (k-slot e of half g <-> accumulator register 8 hf + e)
    f32x16 o[2];
    {
      f16x8 vh8[2 * (AO_RDP + 1)], vl8[2 * (AO_RDP + 1)];   // V fragments of AO_RDP + 1 key steps x 2 d tiles
      auto rd = [&](int ks) {   // key step ks = (kt, hf): both d tiles
        const int kt = ks >> 1, hf = ks & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const unsigned off = AO_VH + dt * 8192 + kt * 2048 + hf * 1024;   // keys 32 kt + 16 hf ..: 256 B per key quad
          const ao_f16x4 h0 = ao_tr(a_v + off), h1 = ao_tr(a_v + off + 512);
          const ao_f16x4 l0 = ao_tr(a_v + off + AO_LO), l1 = ao_tr(a_v + off + 512 + AO_LO);
          vh8[2 * (ks % (AO_RDP + 1)) + dt] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
          vl8[2 * (ks % (AO_RDP + 1)) + dt] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
      };
#pragma unroll
      for (int ks = 0; ks < AO_RDP; ++ks) rd(ks);
      f16x8 ph[2], pl[2];
      {   // the P fragment of key step 0: registers 0 .. 7 of score tile 0 (the later ones are split inside the previous step's statement)
        const float x[8] = {sc[0][0], sc[0][1], sc[0][2], sc[0][3], sc[0][4], sc[0][5], sc[0][6], sc[0][7]};
        ao_split8(x, ph[0], pl[0]);
        asm volatile("s_nop 4" ::"v"(ph[0]), "v"(pl[0]));
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + AO_RDP < 8) rd(ks + AO_RDP);
        const int b = 2 * (ks % (AO_RDP + 1)), kn = (ks + 1) >> 1, hn = (ks + 1) & 1;
        const float x[8] = {sc[kn & 3][hn * 8 + 0], sc[kn & 3][hn * 8 + 1], sc[kn & 3][hn * 8 + 2], sc[kn & 3][hn * 8 + 3],
                            sc[kn & 3][hn * 8 + 4], sc[kn & 3][hn * 8 + 5], sc[kn & 3][hn * 8 + 6], sc[kn & 3][hn * 8 + 7]};
        if (ks == 0) ao_pv_pair<true, true, false>(o[0], o[1], vh8[b], vl8[b], vh8[b + 1], vl8[b + 1], ph[0], pl[0], x, ph[1], pl[1]);
        else if (ks == 7) ao_pv_pair<false, false, true>(o[0], o[1], vh8[b], vl8[b], vh8[b + 1], vl8[b + 1], ph[1], pl[1], x, ph[0], pl[0]);
        else ao_pv_pair<false, true, false>(o[0], o[1], vh8[b], vl8[b], vh8[b + 1], vl8[b + 1], ph[ks & 1], pl[ks & 1], x, ph[(ks + 1) & 1], pl[(ks + 1) & 1]);
      }
    }
    AO_T1(1);
    AO_T0();
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
    AO_T1(2);
    // ---- Bc(h): W_h,0 landed (behind it: W_h,1  W_h,2  Q_h+1 K_h+1); every wave is through with V_h
    ao_sync<4 * AO_UNIT, TM>(&tw[2], &tb[2]);
    if (h < 7) dma_kv(h + 1, 2, AO_VH);
    else dma_prow(0, 8, grpA[1]);
    // ---- the head's four out_proj stages: out^T tile t += Wo[32 t .., k16-step] · O   (hi·hi + hi·lo + lo·hi)
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      if (st == 1) {          // Bd1: W_h,1 landed (behind it: W_h,2  Q K_h+1  V_h+1); stage 0's slot is free -> W_h,3
        ao_sync<4 * AO_UNIT, TM>(&tw[3], &tb[3]);
        dma_w(4 * h + 3, slot);
      } else if (st == 2) {   // Bd2: W_h,2 landed (behind it: Q K_h+1  V_h+1  W_h,3); stage 1's slot -> W_h+1,0
        ao_sync<4 * AO_UNIT, TM>(&tw[4], &tb[4]);
        if (h < 7) dma_w(4 * h + 4, slot == 2 ? 0 : slot + 1);
        else dma_prow(0, 16, grpA[2]);
      } else if (st == 3) {   // Bd3: W_h,3 landed (behind it: W_h+1,0); stage 2's slot -> W_h+1,1
        ao_sync<1 * AO_UNIT, TM>(&tw[5], &tb[5]);
        if (h < 7) dma_w(4 * h + 5, slot == 0 ? 2 : slot - 1);
        else dma_prow(0, 24, grpA[3]);
      }
      AO_T0();
      const int sl_st = st == 0 ? slot : st == 1 ? (slot == 2 ? 0 : slot + 1) : st == 2 ? (slot == 0 ? 2 : slot - 1) : slot;
      const unsigned aw = lds0 + AO_RING + (unsigned)sl_st * AO_SLOT + a_row;
      f16x8 wh[AO_RD + 1], wl[AO_RD + 1];
#pragma unroll
      for (int t = 0; t < AO_RD; ++t) {
        wh[t] = ao_lds128(aw + t * 1024);
        if constexpr (!W2) wl[t] = ao_lds128(aw + t * 1024 + AO_LO);
      }
#pragma unroll
      for (int t = 0; t < AO_NT; ++t) {
        if (t + AO_RD < AO_NT) {
          wh[(t + AO_RD) % (AO_RD + 1)] = ao_lds128(aw + (t + AO_RD) * 1024);
          if constexpr (!W2) wl[(t + AO_RD) % (AO_RD + 1)] = ao_lds128(aw + (t + AO_RD) * 1024 + AO_LO);
        }
        // (asm with an AGPR-pinned accumulator: with builtin MFMAs hipcc moved half of the 15 tiles between the register files
        //  in every head, 450 v_accvgpr_read / _write per iteration)
        if constexpr (W2)
          asm volatile(AO_MFMA "%[c], %[wh], %[oh], %[c]\n\t" AO_MFMA "%[c], %[wh], %[ol], %[c]"
                       : [c] "+a"(pacc[t])
                       : [wh] "v"(wh[t % (AO_RD + 1)]), [oh] "v"(oh[st]), [ol] "v"(ol[st]));
        else
          asm volatile(AO_MFMA "%[c], %[wh], %[oh], %[c]\n\t" AO_MFMA "%[c], %[wh], %[ol], %[c]\n\t" AO_MFMA "%[c], %[wl], %[oh], %[c]"
                       : [c] "+a"(pacc[t])
                       : [wh] "v"(wh[t % (AO_RD + 1)]), [wl] "v"(wl[t % (AO_RD + 1)]), [oh] "v"(oh[st]), [ol] "v"(ol[st]));
      }
      AO_T1(3);
    }
    slot = slot == 2 ? 0 : slot + 1;   // stage 4 (h + 1) = 4 h + 4 -> slot + 4 mod 3
    // ---- Ba(h + 1): Q_h+1, K_h+1 landed (behind them: V_h+1  W_h,3  W_h+1,0  W_h+1,1); stage 3's slot -> W_h+1,2
    ao_sync<4 * AO_UNIT, TM>(&tw[0], &tb[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qh[ks]), "+v"(ql[ks]));
    if (h < 7) dma_w(4 * h + 6, slot == 0 ? 2 : slot - 1);
    else dma_prow(1, 0, grpB0);
    AO_T0();
    if (h < 7) scores();
    AO_T1(4);
  }
  // (the clamped re-loads behind the last head: nothing may land after the workgroup is gone; and the asm MFMAs of the last stage,
  //  which hipcc cannot see, have written their tiles before its v_accvgpr_reads below)
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  if constexpr (TM) t_l1 = __builtin_amdgcn_s_memtime();

  if constexpr (FFN) {
    // ---- Q = P + b_out + out_scale * acc IN REGISTERS (the accumulator layout of out^T — lane (row m, g): columns 8 q + 4 g .. + 3 of every tile — is the
    // layout the FFN stream wants its residual tiles in), tile by tile as whole tuples; the residual rows are where the plain epilogue finds them
    {
      const float* bias = a.bias + g * 4;
      const unsigned gsel = (unsigned)m >> 3;
      const unsigned rowA = (gsel == 0 ? grpA[0] : gsel == 1 ? grpA[1] : gsel == 2 ? grpA[2] : grpA[3]) + ((unsigned)m & 7u) * 1024u - lds0;
      const unsigned rowB = (gsel == 0 ? grpB0 : gsel == 1 ? grpA[0] : gsel == 2 ? grpA[1] : grpA[2]) + ((unsigned)m & 7u) * 1024u - lds0;
#pragma unroll
      for (int t = 0; t < AO_NT; ++t) {
        const int half = t >= 8 ? 1 : 0, tl = t - half * 8;
        if (t == 8) {   // half B, rows 8 .. 31 -> the places of half A's rows 0 .. 23 (this wave's reads of them are complete)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          dma_prow(1, 8, grpA[0]);
          dma_prow(1, 16, grpA[1]);
          dma_prow(1, 24, grpA[2]);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        f32x16 tile = pacc[t];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int col = t * 32 + gq * 8;
          if (col + 8 <= ND) {
            const unsigned c = (unsigned)(tl * 8 + gq * 2) + (unsigned)g;
            const float4 rsd = *reinterpret_cast<const float4*>(smem + (half == 0 ? rowA : rowB) + ((c ^ ((unsigned)m & 15u)) << 4));
            const float4 b = *reinterpret_cast<const float4*>(bias + col);
            tile[gq * 4 + 0] = (tile[gq * 4 + 0] * a.out_scale + b.x) + rsd.x;
            tile[gq * 4 + 1] = (tile[gq * 4 + 1] * a.out_scale + b.y) + rsd.y;
            tile[gq * 4 + 2] = (tile[gq * 4 + 2] * a.out_scale + b.z) + rsd.z;
            tile[gq * 4 + 3] = (tile[gq * 4 + 3] * a.out_scale + b.w) + rsd.w;
          } else {
            tile[gq * 4 + 0] = tile[gq * 4 + 1] = tile[gq * 4 + 2] = tile[gq * 4 + 3] = 0.f;
          }
        }
        asm volatile("" : "+a"(tile));
        pacc[t] = tile;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- everybody is through with the attention's LDS: the FFN ring (2 x 64 KiB at 0) and its tables (behind it) take its place
    constexpr int FP_OFF = 2 * FFN_STAGE, FB2_OFF = FP_OFF + 2 * LN_DP * 4, FB1_OFF = FB2_OFF + 512 * 4;
    static_assert(FB1_OFF + 2048 * 4 <= AO_LDS, "FFN tables inside the kernel's LDS");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      const char* g0 = a.ffn_img + wave * 16384;
#pragma unroll
      for (int k = 0; k < 4; ++k) dma_lin4(voff_lin, g0 + k * 4096, lds0 + wave * 16384 + k * 4096);
    }
    float* sp2 = reinterpret_cast<float*>(smem + FP_OFF);
    float* sb2 = reinterpret_cast<float*>(smem + FB2_OFF);
    float* sb1 = reinterpret_cast<float*>(smem + FB1_OFF);
    for (int i = tid; i < LN_DP; i += 256) {
      const bool in = i < ND;
      sp2[i] = in ? a.ffn_gamma[i] : 0.f;
      sp2[LN_DP + i] = in ? a.ffn_beta[i] : 0.f;
      sb2[i] = in ? a.ffn_b2[i] : 0.f;
    }
    for (int i = tid; i < 2048; i += 256) sb1[i] = i < a.F ? a.ffn_b1[i] : 0.f;
    // two-pass LayerNorm-2 statistics of the lane's row (its half + lane ^ 32) from the tiles
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < AO_NT; ++t) {
      const f32x16 tile = pacc[t];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (t * 4 + (i >> 2) < 58) s1 += tile[i];
      __builtin_amdgcn_sched_barrier(0);
    }
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * (1.0f / 464.0f);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < AO_NT; ++t) {
      const f32x16 tile = pacc[t];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (t * 4 + (i >> 2) < 58) {
          const float d = tile[i] - mean;
          s2 += d * d;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = 1.0f / sqrtf(s2 * (1.0f / 464.0f) + 1e-5f);
    __syncthreads();   // tables visible
    f16x8 xf[29];
    {
      const float* gp = sp2 + g * 4;
      const float* bp = sb2 + g * 4;
#pragma unroll
      for (int t = 0; t < AO_NT; ++t) {
        f32x16 tile = pacc[t];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int gg = t * 4 + gq;
          if (gg < 58) {
            const int ks = gg >> 1, e0 = (gg & 1) * 4;
            const float4 ga = *reinterpret_cast<const float4*>(gp + gg * 8);
            const float4 be = *reinterpret_cast<const float4*>(gp + LN_DP + gg * 8);
            const float4 bb = *reinterpret_cast<const float4*>(bp + gg * 8);
            const float v0 = tile[gq * 4 + 0], v1 = tile[gq * 4 + 1], v2 = tile[gq * 4 + 2], v3 = tile[gq * 4 + 3];
            xf[ks][e0 + 0] = (_Float16)((v0 - mean) * rstd * ga.x + be.x);
            xf[ks][e0 + 1] = (_Float16)((v1 - mean) * rstd * ga.y + be.y);
            xf[ks][e0 + 2] = (_Float16)((v2 - mean) * rstd * ga.z + be.z);
            xf[ks][e0 + 3] = (_Float16)((v3 - mean) * rstd * ga.w + be.w);
            tile[gq * 4 + 0] = v0 + bb.x;
            tile[gq * 4 + 1] = v1 + bb.y;
            tile[gq * 4 + 2] = v2 + bb.z;
            tile[gq * 4 + 3] = v3 + bb.w;
            if (gg & 1) asm volatile("" : "+v"(xf[ks]));
          }
        }
        asm volatile("" : "+a"(tile));
        pacc[t] = tile;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      unsigned relW1[8], relW2[2];
#pragma unroll
      for (int k = 0; k < 8; ++k) relW1[k] = m * RKB + ((((k << 1) | g) ^ (m & 15)) << 4);
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) relW2[sx] = m * 64 + (((2 * sx + g) ^ ((m >> 2) & 3)) << 4);
      const unsigned relB = lds0 + FB1_OFF + g * 16;
      FfnStream<29, AO_NT, 2, false, 6, true> F;
      F.xf = xf;
      F.acc = pacc;
      F.voff = voff_lin;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // chunk 0 (own pieces), then everybody's
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; ++k) F.aW1[k] = lds0 + relW1[k];
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) F.aW2[sx] = lds0 + relW2[sx];
      F.ab_next = relB;
#pragma unroll
      for (int k = 0; k < 29; ++k) asm volatile("" : "+v"(xf[k]));
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_sched_barrier(0);
      F.read_bias();
      F.template prologue<0>();
      {
        const f16x8 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        F.pf[0] = F.pf[1] = F.pfn[0] = F.pfn[1] = z;
      }
      for (int c = 0; c <= a.n_chunks; ++c) {
        F.gnext = a.ffn_img + (size_t)(c == a.n_chunks ? 0 : c + 1) * FFN_STAGE + wave * 16384;
        F.mnext = lds0 + ((c + 1) & 1) * FFN_STAGE + wave * 16384;
        F.ab_next = relB + (c + 1 >= a.n_chunks ? 0 : c + 1) * 128;
        F.template step<0, true>();
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the block's output rows (x + attention + FFN): the layout's real rows only
    if (m < nrow) {
      float* o = a.ffn_out + (row0 + (size_t)(wave * 32 + m)) * ND + g * 4;
#pragma unroll
      for (int t = 0; t < AO_NT; ++t) {
        const f32x16 tile = pacc[t];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int gg = t * 4 + gq;
          if (gg < 58) *reinterpret_cast<float4*>(o + gg * 8) = make_float4(tile[gq * 4 + 0], tile[gq * 4 + 1], tile[gq * 4 + 2], tile[gq * 4 + 3]);
        }
      }
    }
  }
  // ---- epilogue: Q = P + b_out + out_scale * acc.  In accumulator layout a lane owns 16-byte pieces of 32 different rows: a
  // load / store instruction touches 32 cache lines for 1 KiB (first form of this loop: 58 serial round trips, 84 k of the kernel's
  // 263 k cycles; batched four tiles ahead: 60 k — the rate of that access pattern).  So the rows go THROUGH the LDS: residual rows in
  // by LDS-DMA — one instruction per row, whole 128-byte lines —, the arithmetic in place in accumulator layout, rows out as whole lines
  // (one ds_read_b128 + one 1-KiB store per row).  16-byte chunk c of row r sits at chunk c ^ (r & 15): conflict-free for the
  // accumulator-layout accesses (the 16 lanes of a service group are 16 rows with distinct r & 15), linear per row for the DMA and the
  // stores.  Column halves: tiles 0-7 (1 024 B per row), tiles 8-14 (832 B: chunks 0 .. 51).  Half A and rows 0-7 of half B arrived during
  // the last head (dma_prow); rows 8-31 of half B follow into half A's places once its rows have left.  No workgroup barrier: a wave reads
  // only what its own DMA brought and what it wrote itself, in buffers nobody else touches after the last head's barriers.
  if constexpr (!FFN) {
    const float* bias = a.bias + g * 4;
    const unsigned gsel = (unsigned)m >> 3;
    const unsigned rowA = (gsel == 0 ? grpA[0] : gsel == 1 ? grpA[1] : gsel == 2 ? grpA[2] : grpA[3]) + ((unsigned)m & 7u) * 1024u - lds0;   // smem offset of row m
    const unsigned rowB = (gsel == 0 ? grpB0 : gsel == 1 ? grpA[0] : gsel == 2 ? grpA[1] : grpA[2]) + ((unsigned)m & 7u) * 1024u - lds0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      constexpr int kChunks[2] = {64, 52};
      const int nch = kChunks[half];
      if (half == 1) {
        // half B, rows 8 .. 31 -> the places of half A's rows 0 .. 23 (their stores have read them: the lgkmcnt(0) below)
        dma_prow(1, 8, grpA[0]);
        dma_prow(1, 16, grpA[1]);
        dma_prow(1, 24, grpA[2]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // y = acc * out_scale + bias + residual, in place
#pragma unroll
      for (int tl = 0; tl < (half == 0 ? 8 : 7); ++tl)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int t = half * 8 + tl, col = t * 32 + gq * 8;
          if (col + 8 <= ND) {
            const unsigned c = (unsigned)(tl * 8 + gq * 2) + (unsigned)g;        // the lane's chunk of its row m
            float4* pl = reinterpret_cast<float4*>(smem + (half == 0 ? rowA : rowB) + ((c ^ ((unsigned)m & 15u)) << 4));
            const float4 rsd = *pl;
            const float4 b = *reinterpret_cast<const float4*>(bias + col);
            float4 y;
            y.x = (pacc[t][gq * 4 + 0] * a.out_scale + b.x) + rsd.x;
            y.y = (pacc[t][gq * 4 + 1] * a.out_scale + b.y) + rsd.y;
            y.z = (pacc[t][gq * 4 + 2] * a.out_scale + b.z) + rsd.z;
            y.w = (pacc[t][gq * 4 + 3] * a.out_scale + b.w) + rsd.w;
            *pl = y;
          }
        }
      // rows out: physical chunk `lane` of row r is logical chunk lane ^ (r & 15)
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        if (r < nrow) {
          const unsigned lc = lu ^ (unsigned)(r & 15);
          const unsigned base = half == 0 ? grpA[r >> 3] : (r < 8 ? grpB0 : grpA[(r >> 3) - 1]);
          const float4 y = *reinterpret_cast<const float4*>(smem + (base - lds0) + (r & 7) * 1024 + lu * 16u);
          if ((int)lc < nch)
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(a.out + (row0 + (size_t)(wave * 32 + r)) * ND) + half * 1024 + lc * 16) = y;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the rows have left the LDS before the next half's DMA writes it)
    }
  }
  if constexpr (TM) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      for (int i = 0; i < 6; ++i) {
        atomicAdd(&g_attnout_phase[i], tw[i]);
        atomicAdd(&g_attnout_phase[6 + i], tb[i]);
      }
      atomicAdd(&g_attnout_phase[12], t_l1 - t_l0);
      atomicAdd(&g_attnout_phase[13], t_end - t_l1);
      atomicAdd(&g_attnout_phase[14], t_end - t_k0);
      atomicAdd(&g_attnout_phase[15], 1ull);
      for (int i = 0; i < 5; ++i) atomicAdd(&g_attnout_phase[16 + i], tp[i]);
    }
  }
}

#undef AO_T0
#undef AO_T1

void attnout_phase_read(unsigned long long* out24) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_attnout_phase), 24 * sizeof(unsigned long long));
  unsigned long long z[24] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attnout_phase), z, sizeof(z));
}

bool attnout16x3_supported(int S, int H, int dh, int D) { return S >= 1 && S <= 128 && H == 8 && dh == 58 && D == 464; }

int launch_attnout16x3(const AttnOutArgs& a, int B, hipStream_t st) {
  if (!attnout16x3_supported(a.S, 8, 58, a.D) || B < 1 || (a.panel_stride & 15) || !a.qkv_hi || !a.qkv_lo || !a.w_img) return -1;
  // every layout reads 128 rows of 64 bytes from its first row, in every panel: the last one ends 128 - S rows behind the B * S rows in use
  if (a.panel_stride < ((size_t)(B - 1) * a.S + 128) * 64) return -1;
  const bool tm = knob_int("LDM_ATTNOUT_TM", 0) != 0;   // (dev: the phase-timer instantiation)
  if (a.ffn_img && (!a.w2 || a.F < 32 || (a.F & 31) || a.F > 2048 || a.n_chunks != a.F / 32 || !a.ffn_gamma || !a.ffn_beta || !a.ffn_b1 || !a.ffn_b2 || !a.ffn_out))
    return -1;
  auto kern = a.ffn_img ? attnout16x3_k<false, true, true> : a.w2 ? attnout16x3_k<false, true> : tm ? attnout16x3_k<true> : attnout16x3_k<false>;
  allow_big_lds((const void*)kern);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), AO_LDS, st, a);
  return 0;
}

}  // namespace ldm
