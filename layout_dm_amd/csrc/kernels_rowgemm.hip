// Row-stationary MFMA kernels for the K<=512 Linear layers of the denoiser and the fused FFN.
//
// Why: with 128x128 output tiles the K=464 GEMMs move 64 FLOP per byte fetched into a CU and sit
// on the per-CU L2->LDS fill rate (~15-20 B/clk/CU measured, tools/gemm_tune.py).  Here a workgroup
// owns 128 token rows for the WHOLE output width: each of its 4 waves keeps its 32 rows of the
// activation (32 x 512 fp16 = 128 VGPRs) in registers as the MFMA *B* operand for the entire
// kernel and only the weights stream through LDS (LDS-DMA, double buffered), once per workgroup
// => 128 FLOP per fetched byte, and the activation is read from HBM/L2 exactly once.
//
//   rowgemm_k      C[M,N] = epi(A[M,512] · W[N,512]^T + bias)     (QKV, attention out-proj, head)
//   ffn_fused_k    P = Q + relu(H·W1^T + b1)·W2^T + b2            (linear1 -> ReLU -> linear2 +
//                  residual, trainer/models/transformer_utils.py:145-147,179,208-209) — the
//                  [M,1856] hidden activation never leaves the CU: each 32-wide slice of it is
//                  produced in accumulator registers, biased/ReLU'd/cast in-lane and consumed at
//                  once as the B operand of the second GEMM (the k-slot order of that MFMA is
//                  chosen to BE the accumulator layout; W2's K axis is pre-permuted to match).
//
// MFMA form (everything swapped): D[i][j] += A'[i][k] B'[k][j] with A' = weight fragment (i = output
// feature), B' = activation fragment (j = token row) => each lane owns ONE token row and runs of 4
// consecutive output features: 8/16-byte epilogue vectors.
// LDS images are written lane-linearly by the DMA; bank-conflict swizzles are applied to the SOURCE
// address and mirrored on the ds_read_b128 side.
#include <cstdlib>
#include <type_traits>

#include "ldm_kernels.h"
#include "ldm_dma.h"
#include "ldm_pipes.h"

namespace ldm {

typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

__device__ __forceinline__ void dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gas_ptr)g, (las_ptr)l, 16, 0, 0);
}

// One 32-row weight tile (rows n0..n0+31 of W[.,512]) -> LDS stage: instruction i = row i, lane l
// fetches logical 16-B chunk (l ^ (i & 15)) so that physical chunk = logical ^ (row & 15).
__device__ __forceinline__ void issue_w_rows(const __half* W, int row0, char* stage, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = wave + 4 * j;
    dma16(W + (size_t)(row0 + i) * RK + ((lane ^ (i & 15)) << 3), stage + i * RKB);
  }
}

// A' fragment of weight row (tile row r = lane&31) for k16-step ks: logical chunk 2*ks + hi
__device__ __forceinline__ f16x8 read_w_frag(const char* stage, int r, int hi, int ks) {
  return *reinterpret_cast<const f16x8*>(stage + r * RKB + (((2 * ks + hi) ^ (r & 15)) << 4));
}

struct RowEpi {
  const float* bias;
  const float* res;
  float* C32;
  __half* C16;
  int M, N, ldres, ldc32, ldc16, relu;
};

// ---- deferred normalisation helpers -------------------------------------------------------------
// LDS parameter image: sp[0..D) = multiplier (1+scale for AdaLN, gamma for LayerNorm), sp[Dp..Dp+D) = shift/beta
__device__ __forceinline__ void stage_ln_params(float* sp, const LnLoad& ln, int tid) {
  for (int i = tid; i < ln.D; i += 256) {
    sp[i] = ln.ada ? 1.0f + ln.p0[i] : ln.p0[i];
    sp[LN_DP + i] = ln.p1[i];
  }
}
// Builds the register-resident fp16 fragments of token row m from the fp32 row + (mean, rstd):
// y = (x - mean) * rstd * mult + shift   (AdaLayerNorm transformer_utils.py:79-83 / nn.LayerNorm)
template <int KS>
__device__ __forceinline__ void load_xf_ln(f16x8 (&xf)[KS], const LnLoad& ln, int m, int hi, const float* sp) {
  const float2 st = ln.stats[m];
  const float* xr = ln.x + (size_t)m * ln.ldx + hi * 8;
  const float* mp = sp + hi * 8;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float4 a = *reinterpret_cast<const float4*>(xr + ks * 16);
    const float4 b = *reinterpret_cast<const float4*>(xr + ks * 16 + 4);
    const float4 ga = *reinterpret_cast<const float4*>(mp + ks * 16);
    const float4 gb = *reinterpret_cast<const float4*>(mp + ks * 16 + 4);
    const float4 sa = *reinterpret_cast<const float4*>(mp + LN_DP + ks * 16);
    const float4 sb = *reinterpret_cast<const float4*>(mp + LN_DP + ks * 16 + 4);
    xf[ks][0] = (_Float16)fmaf((a.x - st.x) * st.y, ga.x, sa.x);
    xf[ks][1] = (_Float16)fmaf((a.y - st.x) * st.y, ga.y, sa.y);
    xf[ks][2] = (_Float16)fmaf((a.z - st.x) * st.y, ga.z, sa.z);
    xf[ks][3] = (_Float16)fmaf((a.w - st.x) * st.y, ga.w, sa.w);
    xf[ks][4] = (_Float16)fmaf((b.x - st.x) * st.y, gb.x, sb.x);
    xf[ks][5] = (_Float16)fmaf((b.y - st.x) * st.y, gb.y, sb.y);
    xf[ks][6] = (_Float16)fmaf((b.z - st.x) * st.y, gb.z, sb.z);
    xf[ks][7] = (_Float16)fmaf((b.w - st.x) * st.y, gb.w, sb.w);
  }
}
// finish per-row statistics: each lane summed its half of the row, lane^32 holds the other half
__device__ __forceinline__ void store_row_stats(float2* out, int m, int M, int hi, float s1, float s2, int N) {
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  const float mean = s1 / (float)N;
  const float var = fmaxf(s2 / (float)N - mean * mean, 0.f);
  if (hi == 0 && m < M) out[m] = make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
}

// ------------------------------------------------------------------------------------------------
// C[M,N] = epi(A[M,K<=512] · W^T + bias);  KS = number of k16 steps actually used (ceil(K/16)).
// Ordinary global loads inside the tile loop would make hipcc drain the in-flight weight DMA
// (vmcnt(0) at their first use), so: bias lives in LDS, and the residual of tile nt+1 is loaded
// during tile nt and "touched" right after the loop-top vmcnt(0) so no later wait is needed.
template <int KS, int TAG>
__global__ __launch_bounds__(256, 1) void rowgemm_k(const __half* __restrict__ A, const __half* __restrict__ W, int lda,
                                                   int n_tiles, RowEpi e, RowExtra ex) {
  constexpr int TR = 64;               // weight rows per stage (two 32-row MFMA tiles)
  constexpr int STAGE = TR * RKB;      // 64 KiB
  constexpr int PF = 8;                // LDS prefetch depth (fragments in flight ahead of their MFMA)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sbias = reinterpret_cast<float*>(smem + 2 * STAGE);
  float* sp_in = sbias + n_tiles * TR;  // LN-on-load parameters of the A operand
  float* sp_res = sp_in + 2 * LN_DP;   // parameters of the on-the-fly normalised residual
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hi = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + r;
  const bool mok = m < e.M;

  for (int i = tid; i < n_tiles * TR; i += 256) sbias[i] = (e.bias && i < e.N) ? e.bias[i] : 0.f;
  if (ex.in.x) stage_ln_params(sp_in, ex.in, tid);
  if (ex.res.x) stage_ln_params(sp_res, ex.res, tid);
  f16x8 xf[KS];
  if (ex.in.x) {
    __syncthreads();
    load_xf_ln<KS>(xf, ex.in, mok ? m : e.M - 1, hi, sp_in);
  } else {
    const __half* arow = A + (size_t)m * lda + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const f16x8*>(arow + ks * 16);
  }
  float2 rst = make_float2(0.f, 1.f);
  if (ex.res.x) rst = ex.res.stats[mok ? m : e.M - 1];
  float s1 = 0.f, s2 = 0.f;
  const float* rrow = e.res ? e.res + (size_t)(mok ? m : 0) * e.ldres + hi * 4 : nullptr;
  float4 rc[8], rn[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    rc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    rn[g] = rc[g];
    if (rrow && g * 8 + hi * 4 + 3 < e.N) rc[g] = *reinterpret_cast<const float4*>(rrow + g * 8);
  }
  auto issue = [&](int nt, char* stage) {
#pragma unroll
    for (int j = 0; j < TR / 4; ++j) {
      const int i = wave + 4 * j;
      dma16(W + (size_t)(nt * TR + i) * RK + ((lane ^ (i & 15)) << 3), stage + i * RKB);
    }
  };
  issue(0, smem);
  for (int nt = 0; nt < n_tiles; ++nt) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int g = 0; g < 8; ++g) asm volatile("" : "+v"(rc[g].x), "+v"(rc[g].y), "+v"(rc[g].z), "+v"(rc[g].w));
    if (nt + 1 < n_tiles) {
      issue(nt + 1, smem + ((nt + 1) & 1) * STAGE);
      if (rrow) {
#pragma unroll
        for (int g = 0; g < 8; ++g)
          if ((nt + 1) * TR + g * 8 + hi * 4 + 3 < e.N)
            rn[g] = *reinterpret_cast<const float4*>(rrow + (nt + 1) * TR + g * 8);
      }
    }
    const char* st = smem + (nt & 1) * STAGE;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // item it = (ks, t): fragment of weight row t*32 + r, k16-step ks; two independent MFMA chains
    f16x8 q[PF];
#pragma unroll
    for (int it = 0; it < PF; ++it) q[it] = read_w_frag(st + (it & 1) * 32 * RKB, r, hi, it >> 1);
#pragma unroll
    for (int it = 0; it < 2 * KS; ++it) {
      const f16x8 cur = q[it % PF];
      if (it + PF < 2 * KS) q[it % PF] = read_w_frag(st + ((it + PF) & 1) * 32 * RKB, r, hi, (it + PF) >> 1);
      acc[it & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, xf[it >> 1], acc[it & 1], 0, 0, 0);
    }
    if (mok) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int n = nt * TR + g * 8 + hi * 4;
        if (n + 3 < e.N) {
          const float4 b = *reinterpret_cast<const float4*>(sbias + n);
          const int t = g >> 2, rq = g & 3;
          float v0 = acc[t][rq * 4 + 0] + b.x, v1 = acc[t][rq * 4 + 1] + b.y, v2 = acc[t][rq * 4 + 2] + b.z,
                v3 = acc[t][rq * 4 + 3] + b.w;
          if (e.relu) {
            v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
          }
          if (ex.res.x) {  // residual = AdaLN(layer input) recomputed from the raw row + (mean, rstd)
            const float4 gm = *reinterpret_cast<const float4*>(sp_res + n);
            const float4 gs = *reinterpret_cast<const float4*>(sp_res + LN_DP + n);
            v0 += fmaf((rc[g].x - rst.x) * rst.y, gm.x, gs.x);
            v1 += fmaf((rc[g].y - rst.x) * rst.y, gm.y, gs.y);
            v2 += fmaf((rc[g].z - rst.x) * rst.y, gm.z, gs.z);
            v3 += fmaf((rc[g].w - rst.x) * rst.y, gm.w, gs.w);
          } else {
            v0 += rc[g].x; v1 += rc[g].y; v2 += rc[g].z; v3 += rc[g].w;
          }
          s1 += (v0 + v1) + (v2 + v3);
          s2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
          if (e.C32) *reinterpret_cast<float4*>(e.C32 + (size_t)m * e.ldc32 + n) = make_float4(v0, v1, v2, v3);
          if (e.C16) {
            const __half2 h0 = __floats2half2_rn(v0, v1), h1 = __floats2half2_rn(v2, v3);
            uint2 pk;
            pk.x = *reinterpret_cast<const unsigned*>(&h0);
            pk.y = *reinterpret_cast<const unsigned*>(&h1);
            *reinterpret_cast<uint2*>(e.C16 + (size_t)m * e.ldc16 + n) = pk;
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) rc[g] = rn[g];
  }
  if (ex.stats_out) store_row_stats(ex.stats_out, m, e.M, hi, s1, s2, e.N);
}

// ------------------------------------------------------------------------------------------------
// fused FFN, hand-pipelined LDS reads.  hipcc sinks C++-level ds_reads next to their MFMA (read,
// lgkmcnt(0), MFMA — no overlap with ONE wave per SIMD), so every LDS read of the chunk loop is an
// inline-asm ds_read_b128 issued PF items ahead, retired by a counted s_waitcnt lgkmcnt(N) and a
// sched_barrier (cdna guide §5.7, rule 18).  LDS completes in order, so the counts are exact.
// phase-timing instrumentation (LDM_FFN_DBG=3, dev hook only): sums over chunks of s_memtime deltas
__device__ unsigned long long g_ffn_phase[12];

// CH2: GEMM1 runs as TWO interleaved accumulator chains (even / odd k-steps).  A single chain issues a dependent
// v_mfma every step and runs at ~45 cycles per MFMA instead of 32 (profiles/r02_*phase*: gemm1 1350 vs gemm2 969
// cycles per chunk for the same MFMA count, independent of where the weight DMA slots sit); two chains put 64
// cycles between dependent issues.  The bias seeds chain A through the MFMA's C operand (the first MFMA of the
// chunk), so the 16 bias registers die at once and chain B costs no registers over the single-chain form.
template <int KS, int NT2, int PF, int DE = 2, bool TM = false, bool CH2 = false, bool CONSTB = false>
struct FfnPipe {
  static constexpr int NIT = KS + 2 * NT2;
  unsigned long long tC, tD;
  f16x8 q[PF];
  unsigned aW1[8], aW2[2];  // per-lane LDS byte addresses inside the current stage
  const f16x8* xf;
  f32x16 ha;  // (r01: a compiler-scheduled two-chain split pushed the activation fragments into AGPRs and added
              //  ~250 v_accvgpr copies per chunk; the CH2 form below is inline asm with arch-VGPR accumulators)
  f32x16 hb;  // (CH2) second chain
  f32x16* acc;
  f16x8 pf[2];
  float4 bb[4];
  // next chunk's weight DMA (linear 64-KiB image per chunk, 16 KiB per wave): one 1-KiB instruction every
  // DMA_EVERY MFMAs in the MFMA shadow; M0 is rewritten one step before every 4th instruction
  static constexpr int DMA_EVERY = DE;
  static constexpr int IPW = 16;
  const char* gnext;    // image of the next chunk + wave*16 KiB (uniform)
  unsigned mnext;       // LDS byte address of the next stage + wave*16 KiB (uniform)
  unsigned voff;        // lane*16

  template <int J>
  __device__ __forceinline__ void dma_m0() {
    if constexpr (J < IPW && (J & 3) == 0) dma_set_m0(mnext + (J >> 2) * 4096);
  }
  template <int J>
  __device__ __forceinline__ void dma_slot() {
    if constexpr (J < IPW) dma_lin<(J & 3) * 1024>(voff, gnext + (J >> 2) * 4096);
  }

  template <int IT>
  __device__ __forceinline__ void read_item() {
    if constexpr (IT < KS) {
      dsr128<256 * (IT >> 3)>(q[IT % PF], aW1[IT & 7]);
    } else {
      constexpr int sx = (IT - KS) / NT2, t = (IT - KS) % NT2;  // s-major: acc[t] is reused NT2 items later
      dsr128<W1_STAGE + t * 2048>(q[IT % PF], aW2[sx]);
    }
  }
  template <int IT, bool DMA>
  __device__ __forceinline__ void step() {
    if constexpr (IT < NIT) {
      const f16x8 cur_dummy = q[0];
      (void)cur_dummy;
      constexpr int after = (NIT - 1 - IT) < (PF - 1) ? (NIT - 1 - IT) : (PF - 1);  // reads younger than item IT
      wait_lgkm<after>();
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 cur = q[IT % PF];
      if constexpr (CH2 && IT < KS) {
        if constexpr (IT == 0) {
          f32x16 bv;  // bias in accumulator layout = the C operand of chain A's first MFMA
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            bv[rq * 4 + 0] = bb[rq].x; bv[rq * 4 + 1] = bb[rq].y; bv[rq * 4 + 2] = bb[rq].z; bv[rq * 4 + 3] = bb[rq].w;
          }
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(ha) : "v"(cur), "v"(xf[0]), "v"(bv));
        } else if constexpr (IT == 1) {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(hb) : "v"(cur), "v"(xf[CONSTB ? 0 : 1]));
        } else if constexpr (IT % 2 == 0) {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ha) : "v"(cur), "v"(xf[CONSTB ? 0 : IT]));
        } else {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(hb) : "v"(cur), "v"(xf[CONSTB ? 0 : IT]));
        }
        if constexpr (IT == KS - 1) asm volatile("s_nop 15" ::: "memory");
      } else if constexpr (IT == 0) {
        // GEMM1 accumulates in ARCH VGPRs through inline asm: with the builtin hipcc puts `ha` into the AGPR range
        // that the 15 GEMM2 accumulator tiles fill completely and then moves one tile out and back every chunk
        // (32 v_accvgpr copies) plus 16 v_accvgpr_read in the ReLU.  The dependent-MFMA spacing is the same as in
        // the compiler's own code (>= 1 instruction between), the VALU read after the last one gets its wait
        // states from the explicit s_nop below (gfx950: 8 passes + 4 = 12 required).
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ha) : "v"(cur), "v"(xf[0]));
      } else if constexpr (IT < KS) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ha) : "v"(cur), "v"(xf[IT]));
        if constexpr (IT == KS - 1) asm volatile("s_nop 15" ::: "memory");
      } else {
        constexpr int sx = (IT - KS) / NT2, t = (IT - KS) % NT2;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, pf[sx], acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (TM && IT == KS - 1) tC = __builtin_amdgcn_s_memtime();
      if constexpr (TM && IT == KS) tD = __builtin_amdgcn_s_memtime();
      if constexpr (IT + PF < NIT) read_item<IT + PF>();  // refill the slot just consumed
      if constexpr (DMA && IT % DMA_EVERY == DMA_EVERY - 1) dma_slot<IT / DMA_EVERY>();
      else if constexpr (DMA && IT % DMA_EVERY == 0) dma_m0<IT / DMA_EVERY>();
      if constexpr (IT == KS - 1) {
        // bias + ReLU + cast: accumulator reg <-> hidden f = (q&3) + 8*(q>>2) + 4*hi of this chunk
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          if constexpr (CH2) {  // (the bias went in through chain A's C operand)
            pf[rq >> 1][(rq & 1) * 4 + 0] = (_Float16)fmaxf(ha[rq * 4 + 0] + hb[rq * 4 + 0], 0.f);
            pf[rq >> 1][(rq & 1) * 4 + 1] = (_Float16)fmaxf(ha[rq * 4 + 1] + hb[rq * 4 + 1], 0.f);
            pf[rq >> 1][(rq & 1) * 4 + 2] = (_Float16)fmaxf(ha[rq * 4 + 2] + hb[rq * 4 + 2], 0.f);
            pf[rq >> 1][(rq & 1) * 4 + 3] = (_Float16)fmaxf(ha[rq * 4 + 3] + hb[rq * 4 + 3], 0.f);
          } else {
            pf[rq >> 1][(rq & 1) * 4 + 0] = (_Float16)fmaxf(ha[rq * 4 + 0] + bb[rq].x, 0.f);
            pf[rq >> 1][(rq & 1) * 4 + 1] = (_Float16)fmaxf(ha[rq * 4 + 1] + bb[rq].y, 0.f);
            pf[rq >> 1][(rq & 1) * 4 + 2] = (_Float16)fmaxf(ha[rq * 4 + 2] + bb[rq].z, 0.f);
            pf[rq >> 1][(rq & 1) * 4 + 3] = (_Float16)fmaxf(ha[rq * 4 + 3] + bb[rq].w, 0.f);
          }
        }
      }
      step<IT + 1, DMA>();
    }
  }
  template <int IT>
  __device__ __forceinline__ void prologue() {
    if constexpr (IT < PF - 1) {
      read_item<IT>();
      prologue<IT + 1>();
    }
  }
};

// ABL: timing ablations only (1 = no weight DMA after the first chunk, 2 = no LDS reads / MFMAs, 3 = phase timing)
// V  : 0 = r01 prologue / epilogue (loads sunk to their uses: ~6 + 60 dependent memory round trips per block),
//      1 = batched prologue (2 round trips), b2 in LDS, epilogue with the residual loaded 30 column groups at a time
//          (2 round trips), rows >= M written into the padding of `out` (no exec-masked stores); N must be 464.
//      2 = the residual row is read ONCE: the prologue loads x1 in the ACCUMULATOR layout (lane (row, hi) owns
//          columns 8g + 4hi .. +3 of every 8-column group g), seeds the GEMM2 accumulators with x1 + b2 and builds
//          the LN2-normalised fp16 fragments from the same registers — with W1's K axis packed in MFMA k-slot order
//          (ldm_pack::kslot) a lane's accumulator-layout elements of groups 2ks, 2ks+1 ARE its B-operand fragment of
//          k16-step ks.  The epilogue then has no loads at all: statistics + stores.  HBM traffic per block:
//          1 x read + 1 x write of the 128 x 464 fp32 rows instead of 2 x read + 1 x write.  Requires ln.x
//          (deferred normalisation), res == ln.x, and the k-slot W1 image.
// img: per 32-wide hidden chunk c one 64-KiB LDS image (ldm_api.cpp pack_ffn_image):
//   [0, 32 KiB)   W1 rows c*32 .. c*32+31, 1 KiB each, 16-B chunk L of row i at physical chunk L ^ (i & 15)
//   [32, 62 KiB)  W2 (k-slot ordered K axis) columns c*32..+31 of output rows 0..479, 64 B each, chunk L of row n
//                 at physical chunk L ^ ((n >> 2) & 3);   last 2 KiB padding
template <int KS, int NT2, int ABL, int PF = 8, int DE = 2, int V = 0, bool CH2 = false, bool STREAM = false>
__global__ __launch_bounds__(256, 1) void ffn_fused2_k(const __half* __restrict__ H, int ldh, const char* __restrict__ img,
                                                      const float* __restrict__ b1, const float* __restrict__ b2,
                                                      const float* __restrict__ res, float* __restrict__ out, int ldo,
                                                      int M, int N, int n_chunks, LnLoad ln,
                                                      float2* __restrict__ stats_out, int skew) {
  constexpr int STAGE = FFN_STAGE;
  static_assert(W1_STAGE + NT2 * 32 * 64 <= STAGE, "chunk image exceeds its stage");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sb1 = reinterpret_cast<float*>(smem + 2 * STAGE);
  float* sp_in = sb1 + n_chunks * 32;
  float* sb2 = sp_in + 2 * LN_DP;  // (V == 1) linear2 bias, zero beyond N
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hi = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + r;
  // timing ablations (wrong numerics): 4 = phase timing WITHOUT the in-loop weight DMA, 5 = GEMM1 with a CONSTANT B
  // operand (xf[0] in every step), 6 = both
  constexpr bool TM = ABL >= 3 && ABL <= 6;
  unsigned long long t_entry = 0;
  if constexpr (TM) t_entry = __builtin_amdgcn_s_memtime();

  const unsigned lds0 = (unsigned)(size_t)(lds_char_ptr)smem;
  const unsigned voff = lane * 16;
  {  // chunk 0 -> stage 0 (this wave's 16 KiB)
    const char* g0 = img + wave * 16384;
#pragma unroll
    for (int a = 0; a < 4; ++a) dma_lin4(voff, g0 + a * 4096, lds0 + wave * 16384 + a * 4096);
  }
  for (int i = tid; i < n_chunks * 32; i += 256) sb1[i] = b1[i];
  if constexpr (V >= 1)
    for (int i = tid; i < 512; i += 256) sb2[i] = i < N ? b2[i] : 0.f;
  f16x8 xf[KS];
  if (ln.x) {
    stage_ln_params(sp_in, ln, tid);
    __syncthreads();
    if constexpr (V == 1) load_xf_ln_batched<KS, 15>(xf, ln, m < M ? m : M - 1, hi, sp_in);
    else if constexpr (V == 0) load_xf_ln<KS>(xf, ln, m < M ? m : M - 1, hi, sp_in);
  } else {
    const __half* hrow = H + (size_t)m * ldh + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const f16x8*>(hrow + ks * 16);
    // consume the loads here: otherwise hipcc carries "xf may still be in flight" into the chunk loop and puts
    // an s_waitcnt vmcnt(n) in front of every MFMA (it cannot see the explicit waits in the asm statements)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[ks]));
  }
  f32x16 acc[NT2];
#pragma unroll
  for (int t = 0; t < NT2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  if constexpr (V == 2) {
    // one pass over the row in accumulator layout: acc = x1 + b2 (residual + bias seed), xf = fp16(LN2(x1))
    constexpr int NGV = 58, GB = 30;  // 58 valid 8-column groups (N = 464); GB groups (= GB loads) per batch
    const int mr = m < M ? m : M - 1;
    const float2 st = ln.stats[mr];
    const float* xr = ln.x + (size_t)mr * ln.ldx + hi * 4;
    const float* mp = sp_in + hi * 4;
    const float* bp = sb2 + hi * 4;
#pragma unroll
    for (int g0 = 0; g0 < NGV; g0 += GB) {
      float4 raw[GB];
#pragma unroll
      for (int i = 0; i < GB; ++i)
        if (g0 + i < NGV) raw[i] = *reinterpret_cast<const float4*>(xr + (g0 + i) * 8);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        const int gg = g0 + i;
        if (gg < NGV) {
          const float4 a = raw[i];
          const float4 ga = *reinterpret_cast<const float4*>(mp + gg * 8);
          const float4 sa = *reinterpret_cast<const float4*>(mp + LN_DP + gg * 8);
          const float4 bb = *reinterpret_cast<const float4*>(bp + gg * 8);
          const int ks = gg >> 1, e0 = (gg & 1) * 4, t = gg >> 2, q0 = (gg & 3) * 4;
          xf[ks][e0 + 0] = (_Float16)fmaf((a.x - st.x) * st.y, ga.x, sa.x);
          xf[ks][e0 + 1] = (_Float16)fmaf((a.y - st.x) * st.y, ga.y, sa.y);
          xf[ks][e0 + 2] = (_Float16)fmaf((a.z - st.x) * st.y, ga.z, sa.z);
          xf[ks][e0 + 3] = (_Float16)fmaf((a.w - st.x) * st.y, ga.w, sa.w);
          // (the seeds are parked in the AGPR half of the register file right away: with plain assignments hipcc keeps
          //  them in arch VGPRs next to the fragments until the chunk loop and spills ~80 registers to scratch)
          acc[t][q0 + 0] = to_agpr(a.x + bb.x);
          acc[t][q0 + 1] = to_agpr(a.y + bb.y);
          acc[t][q0 + 2] = to_agpr(a.z + bb.z);
          acc[t][q0 + 3] = to_agpr(a.w + bb.w);
          if (gg & 1) asm volatile("" : "+v"(xf[ks]));  // pin (see load_xf_ln_batched)
          if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  unsigned relW1[8], relW2[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) relW1[k] = r * RKB + ((((k << 1) | hi) ^ (r & 15)) << 4);
#pragma unroll
  for (int sx = 0; sx < 2; ++sx) relW2[sx] = r * 64 + (((2 * sx + hi) ^ ((r >> 2) & 3)) << 4);
  const unsigned relB = lds0 + 2 * STAGE + hi * 16;
  // wave skew (experiment): wave w starts every chunk w*skew s_nop-8 later, so that the four waves' 1-KiB DMA
  // instructions do not reach the CU's vector-memory issue port in the same cycle
  const int nskew = wave * skew;

  unsigned long long t_start = 0, t_real0 = 0, s_wait = 0, s_g1 = 0, s_bub = 0, s_g2 = 0;
  if constexpr (TM) {
    t_start = __builtin_amdgcn_s_memtime();
    t_real0 = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (STREAM) {
    static_assert(FFN_STAGE == 0x10000, "stage toggle is address ^ 0x10000");
    FfnStream<KS, NT2, DE, TM> P;
    P.xf = xf;
    P.acc = acc;
    P.voff = voff;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // chunk 0 (own pieces), then everybody's
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) P.aW1[k] = lds0 + relW1[k];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) P.aW2[sx] = lds0 + relW2[sx];
    P.ab_next = relB;
    P.read_bias();
    P.template prologue<0>();
    for (int c = 0; c < n_chunks; ++c) {
      // (the last chunk prefetches chunk 0 again into the idle stage and reads its first fragments: unused)
      P.gnext = img + (size_t)(c + 1 == n_chunks ? 0 : c + 1) * STAGE + wave * 16384;
      P.mnext = lds0 + ((c + 1) & 1) * STAGE + wave * 16384;
      P.ab_next = relB + (c + 1 == n_chunks ? 0 : c + 1) * 128;
      unsigned long long t0 = 0;
      if constexpr (TM) t0 = __builtin_amdgcn_s_memtime();
      P.template step<0, true>();
      if constexpr (TM) {
        const unsigned long long tE = __builtin_amdgcn_s_memtime();
        s_wait += P.tB - P.tA;
        s_g1 += P.tC - t0;
        s_bub += P.tD - P.tC;
        s_g2 += (tE - P.tD) - (P.tB - P.tA);
      }
    }
  } else {
  FfnPipe<KS, NT2, PF, DE, TM, CH2, (ABL == 5 || ABL == 6)> P;
  P.xf = xf;
  P.acc = acc;
  P.voff = voff;
  auto chunk = [&](int c, auto dma_tag) {
    constexpr bool DMA = decltype(dma_tag)::value;
    unsigned long long tA = 0, tB = 0;
    if constexpr (TM) tA = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int i = 0; i < nskew; ++i) asm volatile("s_nop 7");
    if constexpr (TM) {
      tB = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    P.gnext = img + (size_t)((ABL == 1 || c + 1 == n_chunks) ? 0 : c + 1) * STAGE + wave * 16384;
    P.mnext = lds0 + ((c + 1) & 1) * STAGE + wave * 16384;
    const unsigned sbase = lds0 + (c & 1) * STAGE;
#pragma unroll
    for (int k = 0; k < 8; ++k) P.aW1[k] = sbase + relW1[k];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) P.aW2[sx] = sbase + relW2[sx];
    const unsigned ab = relB + c * 128;  // b1[c*32 + rq*8 + hi*4 ..]
    dsr128f<0>(P.bb[0], ab);
    dsr128f<32>(P.bb[1], ab);
    dsr128f<64>(P.bb[2], ab);
    dsr128f<96>(P.bb[3], ab);
    if constexpr (ABL == 2) {
      if constexpr (DMA) {
#pragma unroll
        for (int a = 0; a < 4; ++a) dma_lin4(voff, P.gnext + a * 4096, P.mnext + a * 4096);
      }
    } else {
      P.template prologue<0>();
      // queue holds items 0..PF-2; step<IT> waits for item IT, runs its MFMA, then issues item IT+PF
      P.template read_item<PF - 1>();
      P.template step<0, DMA>();
    }
    if constexpr (TM) {
      const unsigned long long tE = __builtin_amdgcn_s_memtime();
      s_wait += tB - tA;
      s_g1 += P.tC - tB;
      s_bub += P.tD - P.tC;
      s_g2 += tE - P.tD;
    }
  };
  // (a peeled DMA-free last chunk made hipcc spill MFMA operands in the peeled copy: the last chunk simply
  //  prefetches chunk 0 again into the idle stage — 1/58 extra L2 reads, drained below)
  for (int c = 0; c < n_chunks; ++c) chunk(c, std::integral_constant<bool, ABL != 4 && ABL != 6>{});
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t_end = 0, t_real1 = 0;
  if constexpr (TM) {
    t_end = __builtin_amdgcn_s_memtime();
    t_real1 = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (V == 2) {
    // the accumulators were seeded with x1 + b2: statistics + stores only (rows >= M land in the padding of `out`)
    int me = m, hie = hi;
    asm volatile("" : "+v"(me), "+v"(hie));
    float* orow = out + (size_t)me * ldo + hie * 4;
    constexpr int NGV = 58;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int gg = 0; gg < NGV; ++gg) {
      const int t = gg >> 2, rq = gg & 3;
      const float v0 = acc[t][rq * 4 + 0], v1 = acc[t][rq * 4 + 1], v2 = acc[t][rq * 4 + 2], v3 = acc[t][rq * 4 + 3];
      s1 += (v0 + v1) + (v2 + v3);
      s2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      *reinterpret_cast<float4*>(orow + gg * 8) = make_float4(v0, v1, v2, v3);
    }
    if (stats_out) store_row_stats(stats_out, m, M, hi, s1, s2, N);
  } else if constexpr (V == 1) {
    // residual + bias + row statistics, the residual row pieces fetched GB column groups at a time (the chunk
    // loop's 116 fragment registers are dead here).  The row / lane-half are re-materialised behind an opaque
    // asm so that hipcc cannot hoist the addresses (and then the loads) above the last chunk.
    int me = m, hie = hi;
    asm volatile("" : "+v"(me), "+v"(hie));
    const int mr = me < M ? me : M - 1;
    const float* rrow = res + (size_t)mr * ldo + hie * 4;
    float* orow = out + (size_t)me * ldo + hie * 4;  // rows >= M: padding rows of `out` (allocated by the host)
    const float* brow = sb2 + hie * 4;
    constexpr int NGV = 58;  // valid 8-column groups: N = 464 (checked by the launcher)
    constexpr int GB = 30;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      float4 rv[GB];
#pragma unroll
      for (int g = 0; g < GB; ++g)
        if (h2 * GB + g < NGV) rv[g] = *reinterpret_cast<const float4*>(rrow + (h2 * GB + g) * 8);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < GB; ++g) {
        const int gg = h2 * GB + g;
        if (gg < NGV) {
          const int t = gg >> 2, rq = gg & 3;
          const float4 b = *reinterpret_cast<const float4*>(brow + gg * 8);
          const float v0 = acc[t][rq * 4 + 0] + b.x + rv[g].x, v1 = acc[t][rq * 4 + 1] + b.y + rv[g].y;
          const float v2 = acc[t][rq * 4 + 2] + b.z + rv[g].z, v3 = acc[t][rq * 4 + 3] + b.w + rv[g].w;
          s1 += (v0 + v1) + (v2 + v3);
          s2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
          *reinterpret_cast<float4*>(orow + gg * 8) = make_float4(v0, v1, v2, v3);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (stats_out) store_row_stats(stats_out, m, M, hi, s1, s2, N);
  } else {
    // row / lane-half re-materialised behind an opaque asm: hipcc otherwise hoists the 60 epilogue addresses and
    // masks above the last chunk and spills MFMA operands to scratch there
    int me = m, hie = hi;
    asm volatile("" : "+v"(me), "+v"(hie));
    const int mr = me < M ? me : M - 1;
    const float* rrow = res + (size_t)mr * ldo + hie * 4;
    float* orow = out + (size_t)mr * ldo + hie * 4;
    const float* brow = b2 + hie * 4;
    const bool mok = me < M;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int n0 = t * 32 + rq * 8;  // N % 8 == 0: the group is inside or outside for both lane halves
        if (n0 + 8 <= N) {
          const float4 b = *reinterpret_cast<const float4*>(brow + n0);
          const float4 qv = *reinterpret_cast<const float4*>(rrow + n0);
          const float v0 = acc[t][rq * 4 + 0] + b.x + qv.x, v1 = acc[t][rq * 4 + 1] + b.y + qv.y;
          const float v2 = acc[t][rq * 4 + 2] + b.z + qv.z, v3 = acc[t][rq * 4 + 3] + b.w + qv.w;
          s1 += (v0 + v1) + (v2 + v3);
          s2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
          if (mok) *reinterpret_cast<float4*>(orow + n0) = make_float4(v0, v1, v2, v3);
        }
      }
    }
    if (stats_out) store_row_stats(stats_out, m, M, hi, s1, s2, N);
  }
  if constexpr (TM) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the epilogue time includes the drain of its stores
    const unsigned long long t_exit = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      atomicAdd(&g_ffn_phase[0], 1ull);
      atomicAdd(&g_ffn_phase[1], t_end - t_start);
      atomicAdd(&g_ffn_phase[2], t_real1 - t_real0);
      atomicAdd(&g_ffn_phase[3], s_wait);
      atomicAdd(&g_ffn_phase[4], s_g1);
      atomicAdd(&g_ffn_phase[5], s_bub);
      atomicAdd(&g_ffn_phase[6], s_g2);
      atomicAdd(&g_ffn_phase[7], t_start - t_entry);  // prologue: LN parameters, row loads, fragments
      atomicAdd(&g_ffn_phase[8], t_exit - t_end);     // epilogue: residual, bias, statistics, stores
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Hand-pipelined row-stationary GEMM with fp16 output (QKV projection): C16[M,N] = A·W^T + bias.
// 64 weight rows per LDS stage = two 32-row MFMA tiles = two independent accumulator chains; LDS reads
// by inline asm PF items ahead; the next tile's DMA is interleaved (one 1-KiB instruction every 2
// MFMAs); the tile's 8 stores stay in flight across the barrier (loop-top wait is vmcnt(8): DMA is
// older than the stores in the in-order vmcnt queue).  Rows >= M are written too (buffers are padded).
template <int KS, int PF, int NT = 2>
struct RowPipe {
  static constexpr int NIT = NT * KS;
  f16x8 q[PF];
  unsigned aW[8];
  const f16x8* xf;
  f32x16 acc0, acc1;
  const char* gW;   // next tile's weight rows (uniform)
  char* nstage;     // LDS stage of the next tile (uniform)
  unsigned lo1[4];
  int wave;
  bool has_next;

  template <int IT>
  __device__ __forceinline__ void read_item() {
    constexpr int t = IT % NT, ks = IT / NT;
    dsr128<t * 32 * RKB + 256 * (ks >> 3)>(q[IT % PF], aW[ks & 7]);
  }
  template <int J>
  __device__ __forceinline__ void dma_slot() {
    if constexpr (J < 8 * NT) {
      if (has_next) {
        const int i = wave + 4 * J;
        dma16(gW + i * RKB + lo1[J & 3], nstage + i * RKB);
      }
    }
  }
  template <int IT>
  __device__ __forceinline__ void step() {
    if constexpr (IT < NIT) {
      constexpr int after = (NIT - 1 - IT) < (PF - 1) ? (NIT - 1 - IT) : (PF - 1);
      wait_lgkm<after>();
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 cur = q[IT % PF];
      if constexpr (IT % NT == 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, xf[IT / NT], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, xf[IT / NT], acc0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (IT + PF < NIT) read_item<IT + PF>();
      if constexpr (IT % 2 == 1) dma_slot<IT / 2>();
      step<IT + 1>();
    }
  }
  template <int IT>
  __device__ __forceinline__ void prologue() {
    if constexpr (IT < PF) {
      read_item<IT>();
      prologue<IT + 1>();
    }
  }
};

template <int KS, int TR>
__global__ __launch_bounds__(256, 1) void rowgemm16_k(const __half* __restrict__ A, const __half* __restrict__ W, int lda,
                                                     int n_tiles, const float* __restrict__ bias, __half* __restrict__ C16,
                                                     int ldc, int N, int M, LnLoad ln) {
  constexpr int NT = TR / 32;
  constexpr int STAGE = TR * RKB;
  constexpr int PF = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sbias = reinterpret_cast<float*>(smem + 2 * STAGE);
  float* sp_in = sbias + n_tiles * TR;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hi = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + r;

  for (int i = tid; i < n_tiles * TR; i += 256) sbias[i] = (bias && i < N) ? bias[i] : 0.f;
  f16x8 xf[KS];
  if (ln.x) {
    stage_ln_params(sp_in, ln, tid);
    __syncthreads();
    load_xf_ln<KS>(xf, ln, m < M ? m : M - 1, hi, sp_in);
  } else {
    const __half* arow = A + (size_t)m * lda + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const f16x8*>(arow + ks * 16);
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_char_ptr)smem;
  unsigned relW[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) relW[k] = r * RKB + ((((k << 1) | hi) ^ (r & 15)) << 4);
  RowPipe<KS, PF, NT> P;
  P.xf = xf;
  P.wave = wave;
#pragma unroll
  for (int jm = 0; jm < 4; ++jm) P.lo1[jm] = (unsigned)((lane ^ ((wave + 4 * jm) & 15)) << 4);
  {  // first tile: burst
#pragma unroll
    for (int j = 0; j < 8 * NT; ++j) {
      const int i = wave + 4 * j;
      dma16(W + (size_t)i * RK + ((lane ^ (i & 15)) << 3), smem + i * RKB);
    }
  }
  __half* crow = C16 + (size_t)m * ldc + hi * 4;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  for (int nt = 0; nt < n_tiles; ++nt) {
    // outstanding VMEM, oldest first: [DMA of this tile] [stores of the previous tile x 4*NT]
    if (nt > 0) {
      if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    P.has_next = nt + 1 < n_tiles;
    P.gW = reinterpret_cast<const char*>(W + (size_t)(nt + 1) * TR * RK);
    P.nstage = smem + ((nt + 1) & 1) * STAGE;
    const unsigned sbase = lds0 + (nt & 1) * STAGE;
#pragma unroll
    for (int k = 0; k < 8; ++k) P.aW[k] = sbase + relW[k];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      P.acc0[i] = 0.f;
      P.acc1[i] = 0.f;
    }
    P.template prologue<0>();
    P.template step<0>();
    // epilogue: bias + cast, 8-byte stores (always issued: exactly 4*NT VMEM ops per tile per wave)
#pragma unroll
    for (int g = 0; g < 4 * NT; ++g) {
      const int n = nt * TR + g * 8 + hi * 4;
      const float4 b = *reinterpret_cast<const float4*>(sbias + n);
      const int rq = g & 3;
      const float v0 = (g < 4 ? P.acc0[rq * 4 + 0] : P.acc1[rq * 4 + 0]) + b.x;
      const float v1 = (g < 4 ? P.acc0[rq * 4 + 1] : P.acc1[rq * 4 + 1]) + b.y;
      const float v2 = (g < 4 ? P.acc0[rq * 4 + 2] : P.acc1[rq * 4 + 2]) + b.z;
      const float v3 = (g < 4 ? P.acc0[rq * 4 + 3] : P.acc1[rq * 4 + 3]) + b.w;
      const __half2 h0 = __floats2half2_rn(v0, v1), h1 = __floats2half2_rn(v2, v3);
      uint2 pk;
      pk.x = *reinterpret_cast<const unsigned*>(&h0);
      pk.y = *reinterpret_cast<const unsigned*>(&h1);
      *reinterpret_cast<uint2*>(crow + nt * TR + g * 8) = pk;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Hand-pipelined row-stationary GEMM with fp32 output + residual + row statistics (attention
// out-projection):  C32 = A·W^T + bias + LN(res) ; stats_out = (mean, rstd) of every output row.
// Same pipeline as rowgemm16_k; the tile's residual rows are fetched right after the barrier (ordinary
// loads, fenced by a sched_barrier so they are not sunk) and have the whole MFMA phase to land.
template <int KS, int TR>
__global__ __launch_bounds__(256, 1) void rowgemm32_k(const __half* __restrict__ A, int lda, const __half* __restrict__ W,
                                                     int n_tiles, const float* __restrict__ bias, float* __restrict__ C32,
                                                     int ldc, int N, int M, LnLoad rs, float2* __restrict__ stats_out) {
  constexpr int NT = TR / 32;
  constexpr int STAGE = TR * RKB;
  constexpr int PF = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sbias = reinterpret_cast<float*>(smem + 2 * STAGE);
  float* sp_res = sbias + n_tiles * TR;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hi = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + r;
  const int mr = m < M ? m : M - 1;

  for (int i = tid; i < n_tiles * TR; i += 256) sbias[i] = (bias && i < N) ? bias[i] : 0.f;
  stage_ln_params(sp_res, rs, tid);
  f16x8 xf[KS];
  {
    const __half* arow = A + (size_t)m * lda + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const f16x8*>(arow + ks * 16);
  }
  const float2 rst = rs.stats[mr];
  const float* rrow = rs.x + (size_t)mr * rs.ldx + hi * 4;
  const unsigned lds0 = (unsigned)(size_t)(lds_char_ptr)smem;
  unsigned relW[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) relW[k] = r * RKB + ((((k << 1) | hi) ^ (r & 15)) << 4);
  RowPipe<KS, PF, NT> P;
  P.xf = xf;
  P.wave = wave;
#pragma unroll
  for (int jm = 0; jm < 4; ++jm) P.lo1[jm] = (unsigned)((lane ^ ((wave + 4 * jm) & 15)) << 4);
#pragma unroll
  for (int j = 0; j < 8 * NT; ++j) {
    const int i = wave + 4 * j;
    dma16(W + (size_t)i * RK + ((lane ^ (i & 15)) << 3), smem + i * RKB);
  }
  float* crow = C32 + (size_t)m * ldc + hi * 4;
  float s1 = 0.f, s2 = 0.f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  for (int nt = 0; nt < n_tiles; ++nt) {
    // outstanding VMEM, oldest first: [DMA of this tile] [stores of the previous tile x 4*NT]
    if (nt > 0) {
      if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float4 rv[4 * NT];
#pragma unroll
    for (int g = 0; g < 4 * NT; ++g) {
      const int n = nt * TR + g * 8 + hi * 4;
      rv[g] = (n + 3 < N) ? *reinterpret_cast<const float4*>(rrow + nt * TR + g * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);
    P.has_next = nt + 1 < n_tiles;
    P.gW = reinterpret_cast<const char*>(W + (size_t)(nt + 1) * TR * RK);
    P.nstage = smem + ((nt + 1) & 1) * STAGE;
    const unsigned sbase = lds0 + (nt & 1) * STAGE;
#pragma unroll
    for (int k = 0; k < 8; ++k) P.aW[k] = sbase + relW[k];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      P.acc0[i] = 0.f;
      P.acc1[i] = 0.f;
    }
    P.template prologue<0>();
    P.template step<0>();
#pragma unroll
    for (int g = 0; g < 4 * NT; ++g) {
      const int n = nt * TR + g * 8 + hi * 4;
      const bool ok = n + 3 < N;
      const float4 b = *reinterpret_cast<const float4*>(sbias + n);
      const float4 gm = *reinterpret_cast<const float4*>(sp_res + (ok ? n : 0));
      const float4 gs = *reinterpret_cast<const float4*>(sp_res + LN_DP + (ok ? n : 0));
      const int rq = g & 3;
      float v0 = (g < 4 ? P.acc0[rq * 4 + 0] : P.acc1[rq * 4 + 0]) + b.x + fmaf((rv[g].x - rst.x) * rst.y, gm.x, gs.x);
      float v1 = (g < 4 ? P.acc0[rq * 4 + 1] : P.acc1[rq * 4 + 1]) + b.y + fmaf((rv[g].y - rst.x) * rst.y, gm.y, gs.y);
      float v2 = (g < 4 ? P.acc0[rq * 4 + 2] : P.acc1[rq * 4 + 2]) + b.z + fmaf((rv[g].z - rst.x) * rst.y, gm.z, gs.z);
      float v3 = (g < 4 ? P.acc0[rq * 4 + 3] : P.acc1[rq * 4 + 3]) + b.w + fmaf((rv[g].w - rst.x) * rst.y, gm.w, gs.w);
      if (!ok) { v0 = 0.f; v1 = 0.f; v2 = 0.f; v3 = 0.f; }
      s1 += (v0 + v1) + (v2 + v3);
      s2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      // a column group is invalid for ALL lanes or none and only in the LAST tile, so every tile that is
      // followed by a loop-top vmcnt wait issues exactly 4*NT store instructions per wave
      if (ok) *reinterpret_cast<float4*>(crow + nt * TR + g * 8) = make_float4(v0, v1, v2, v3);
    }
  }
  if (stats_out) store_row_stats(stats_out, m, M, hi, s1, s2, N);
}

// ------------------------------------------------------------------------------------------------
template <int KS, int TAG>
static void launch_rowgemm_t(const GemmArgs& g, const RowExtra& ex, hipStream_t st) {
  RowEpi e{g.bias, g.res, g.C32, g.C16, g.M, g.N, g.ldres, g.ldc32, g.ldc16, g.relu};
  const int n_tiles = (g.N + 63) / 64;
  const int lds = 2 * 64 * RKB + n_tiles * 64 * 4 + 4 * LN_DP * 4;
  auto kern = rowgemm_k<KS, TAG>;
  allow_big_lds((const void*)kern);
  hipLaunchKernelGGL(kern, dim3((g.M + 127) / 128), dim3(256), lds, st, (const __half*)g.A, (const __half*)g.W, g.lda,
                     n_tiles, e, ex);
}

// A: [>=ceil(M/128)*128 rows, lda] fp16 with K zero-padded to a multiple of 16 (<=512);
// W: [>=ceil(N/64)*64 rows, 512] fp16.  N must be a multiple of 4.
void launch_rowgemm(const GemmArgs& g, int tag, const RowExtra* exp, hipStream_t st) {
  const bool k29 = g.K <= 464;
  RowExtra ex{};
  if (exp) ex = *exp;
  if (tag == 0 && k29 && g.C16 && !g.C32 && !g.res && !g.relu && g.N % 64 == 0 && !ex.res.x && !ex.stats_out &&
      !getenv("LDM_ROWGEMM_V1")) {
    constexpr int KS = 29;
    static const int tr = getenv("LDM_ROW_TR") ? atoi(getenv("LDM_ROW_TR")) : 32;
    const int n_tiles = g.N / tr;
    const int lds = 2 * tr * RKB + g.N * 4 + 2 * LN_DP * 4;
    auto kern = tr == 32 ? rowgemm16_k<KS, 32> : rowgemm16_k<KS, 64>;
    allow_big_lds((const void*)kern);
    hipLaunchKernelGGL(kern, dim3((g.M + 127) / 128), dim3(256), lds, st, (const __half*)g.A, (const __half*)g.W,
                       g.lda, n_tiles, g.bias, g.C16, g.ldc16, g.N, g.M, ex.in);
    return;
  }
  if (tag == 1 && !k29 && g.K <= 512 && g.C32 && !g.C16 && !g.relu && ex.res.x && !ex.in.x && g.ldc32 % 4 == 0 &&
      g.N % 8 == 0 && !getenv("LDM_ROWGEMM_V1")) {
    constexpr int KS = 32;
    static const int tr = getenv("LDM_ROW_TR") ? atoi(getenv("LDM_ROW_TR")) : 32;
    const int n_tiles = (g.N + tr - 1) / tr;
    const int lds = 2 * tr * RKB + n_tiles * tr * 4 + 2 * LN_DP * 4;
    auto kern = tr == 32 ? rowgemm32_k<KS, 32> : rowgemm32_k<KS, 64>;
    allow_big_lds((const void*)kern);
    hipLaunchKernelGGL(kern, dim3((g.M + 127) / 128), dim3(256), lds, st, (const __half*)g.A, g.lda, (const __half*)g.W,
                       n_tiles, g.bias, g.C32, g.ldc32, g.N, g.M, ex.res, ex.stats_out);
    return;
  }
  switch (tag) {
    case 0: k29 ? launch_rowgemm_t<29, 0>(g, ex, st) : launch_rowgemm_t<32, 0>(g, ex, st); break;
    case 1: k29 ? launch_rowgemm_t<29, 1>(g, ex, st) : launch_rowgemm_t<32, 1>(g, ex, st); break;
    case 2: k29 ? launch_rowgemm_t<29, 2>(g, ex, st) : launch_rowgemm_t<32, 2>(g, ex, st); break;
    default: k29 ? launch_rowgemm_t<29, 4>(g, ex, st) : launch_rowgemm_t<32, 4>(g, ex, st); break;
  }
}

// LDM_FFN_V (latched): 0 = r01 prologue / epilogue, 1 = batched, 2 = single read of the residual row (default)
int ffn_fused_version() {
  static const int v = getenv("LDM_FFN_V") ? atoi(getenv("LDM_FFN_V")) : 2;
  return v;
}

// img: the W1 | W2 LDS image; for version 2 (img_ks != nullptr is then required) W1's K axis is k-slot ordered.
void launch_ffn_fused(const __half* H, int ldh, const void* img, const void* img_ks, const float* b1, const float* b2,
                      const float* res, float* out, int ldo, int M, int N, int F, const LnLoad* lnp, float2* stats_out,
                      hipStream_t st) {
  LnLoad ln{};
  if (lnp) ln = *lnp;
  constexpr int NT2 = 15, KS = 29;  // N <= 480, K <= 464 (d_model 464 = 29 x 16)
  const int lds = 2 * FFN_STAGE + F * 4 + 2 * LN_DP * 4 + 512 * 4;
  static const int dbg = getenv("LDM_FFN_DBG") ? atoi(getenv("LDM_FFN_DBG")) : 0;
  static const int var = getenv("LDM_FFN_VAR") ? atoi(getenv("LDM_FFN_VAR")) : 0;
  static const int skew = getenv("LDM_FFN_SKEW") ? atoi(getenv("LDM_FFN_SKEW")) : 0;
  static const int ch = getenv("LDM_FFN_CH") ? atoi(getenv("LDM_FFN_CH")) : 2;  // GEMM1 accumulator chains (version 2)
  static const int stream = getenv("LDM_FFN_STREAM") ? atoi(getenv("LDM_FFN_STREAM")) : 1;  // continuous pipeline
  // versions 1 / 2 need N == 464 and `out` padded to a multiple of 128 rows (the engine's workspace is); version 2
  // additionally the deferred-normalisation input with res == ln.x and the k-slot image
  int ver = (N == 464) ? ffn_fused_version() : 0;
  if (ver == 2 && !(ln.x && ln.x == res && img_ks && ln.ldx == ldo)) ver = 1;
  using K = void (*)(const __half*, int, const char*, const float*, const float*, const float*, float*, int, int, int,
                     int, LnLoad, float2*, int);
  K kern;
  if (ver == 0)
    kern = dbg == 1   ? ffn_fused2_k<KS, NT2, 1>
           : dbg == 2 ? ffn_fused2_k<KS, NT2, 2>
           : dbg == 3 ? ffn_fused2_k<KS, NT2, 3>
                      : ffn_fused2_k<KS, NT2, 0>;
  else if (ver == 1)
    kern = dbg == 3   ? ffn_fused2_k<KS, NT2, 3, 8, 2, 1>
           : var == 3 ? ffn_fused2_k<KS, NT2, 0, 8, 3, 1>
                      : ffn_fused2_k<KS, NT2, 0, 8, 2, 1>;
  else
    kern = (stream && dbg == 0 && ch == 2) ? ffn_fused2_k<KS, NT2, 0, 8, 2, 2, true, true>
           : (stream && dbg == 3 && ch == 2) ? ffn_fused2_k<KS, NT2, 3, 8, 2, 2, true, true>
           : dbg == 4 ? ffn_fused2_k<KS, NT2, 4, 8, 2, 2, true>
           : dbg == 5 ? ffn_fused2_k<KS, NT2, 5, 8, 2, 2, true>
           : dbg == 6 ? ffn_fused2_k<KS, NT2, 6, 8, 2, 2, true>
           : ch == 1 ? (dbg == 3 ? ffn_fused2_k<KS, NT2, 3, 8, 2, 2> : ffn_fused2_k<KS, NT2, 0, 8, 2, 2>)
                     : (dbg == 3 ? ffn_fused2_k<KS, NT2, 3, 8, 2, 2, true> : ffn_fused2_k<KS, NT2, 0, 8, 2, 2, true>);
  allow_big_lds((const void*)kern);
  hipLaunchKernelGGL(kern, dim3((M + 127) / 128), dim3(256), lds, st, H, ldh, (const char*)(ver == 2 ? img_ks : img), b1,
                     b2, res, out, ldo, M, N, F / 32, ln, stats_out, skew);
}

// dev hook: read + reset the phase sums {blocks, cycles, realtime ticks, wait, gemm1, bubble, gemm2, -}
void ffn_phase_read(unsigned long long* out12) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_ffn_phase), 12 * sizeof(unsigned long long));
  unsigned long long z[12] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ffn_phase), z, sizeof(z));
}

}  // namespace ldm
