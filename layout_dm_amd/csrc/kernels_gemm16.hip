// fp16-operand / fp32-accumulate GEMM for the fast numerics mode:
//     C[M,N] = epilogue(A[M,K] · W[N,K]^T + bias[N])
// (torch.nn.Linear sites of the denoiser: trainer/models/transformer_utils.py:140-147,197-209,
//  trainer/models/common/nn_lib.py:186-189).
//
// gfx950 structure (cdna_hip_programming.md §5):
//  * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no
//    VGPR staging); NSTAGE LDS ring, tiles kt+1..kt+NSTAGE-2 stay in flight ACROSS the single
//    s_barrier per K-tile (counted s_waitcnt vmcnt(N), never 0 in steady state).
//  * the DMA writes LDS lane-linearly, so the bank-conflict swizzle is applied to the SOURCE
//    address (16-B chunk index XOR f(row)) and again on the ds_read_b128 side (rule 21).
//  * v_mfma_f32_32x32x16_f16 with SWAPPED operands (first operand = W fragment): the accumulator
//    tile is D[i = n][j = m], so each lane owns ONE output row m and runs of 4 consecutive n —
//    bias / residual / stores are 8- or 16-byte vector accesses instead of 2-byte scatters.
//  * XCD-aware tile order (8 private L2s; block b runs on XCD b%8).
//
// Buffers are padded by the host so no load needs a bounds check: A has >= ceil(M/BM)*BM rows,
// W has >= ceil(N/BN)*BN rows (zero rows), K is a multiple of BK with zero pad columns.
#include "ldm_kernels.h"

namespace ldm {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

__device__ __forceinline__ int xcd_remap16(int bid, int nwg) {
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Epi16 {
  const float* bias;
  const float* res;
  float* C32;
  __half* C16;
  int M, N, ldres, ldc32, ldc16, relu;
};

// TAG only names the call site (0 qkv, 1 attn_out, 2 ffn1, 3 ffn2, 4 head) so that rocprofv3 reports
// the five Linear classes as separate kernels.
template <int BM, int BN, int BK, int NSTAGE, int WM, int WN, int TAG>
__global__ __launch_bounds__(WM* WN * 64) void gemm16_k(const __half* __restrict__ A, const __half* __restrict__ W,
                                                         int lda, int ldw, int K, int tiles_n, Epi16 e) {
  constexpr int NW = WM * WN;
  constexpr int RB = BK * 2;          // bytes per tile row
  constexpr int CPR = RB / 16;        // 16-B chunks per row (4 or 8)
  constexpr int RPI = 64 / CPR;       // rows per 1-KiB DMA instruction
  constexpr int NINST = (BM + BN) / RPI;
  constexpr int LPT = NINST / NW;     // DMA instructions per wave per tile
  static_assert(NINST % NW == 0, "tile does not split evenly over the waves");
  constexpr int STAGE_BYTES = (BM + BN) * RB;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // 32x32 accumulators per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap16(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * BM;
  const int n0 = (tile % tiles_n) * BN;

  // ---- DMA source pointers (per lane), one per instruction slot of this wave
  const int lrow = lane / CPR;
  const int lchunk = lane % CPR;
  const __half* src[LPT];
#pragma unroll
  for (int j = 0; j < LPT; ++j) {
    const int inst = wave + j * NW;
    const int R = inst * RPI + lrow;  // row in the concatenated [A rows ; W rows] tile
    const bool isA = (inst * RPI) < BM;
    const int rt = isA ? R : R - BM;
    const int sw = (CPR == 8) ? ((rt >> 1) & 7) : ((rt >> 2) & 3);
    const int c = lchunk ^ sw;
    src[j] = isA ? A + (size_t)(m0 + rt) * lda + c * 8 : W + (size_t)(n0 + rt) * ldw + c * 8;
  }
  auto issue = [&](int kt, int stage) {
    char* sbase = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      const int inst = wave + j * NW;
      __builtin_amdgcn_global_load_lds((gas_ptr)(src[j] + (size_t)kt * BK), (las_ptr)(sbase + inst * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes inside a stage)
  const int frow = lane & 31;
  const int hi = lane >> 5;
  const int fsw = (CPR == 8) ? ((frow >> 1) & 7) : ((frow >> 2) & 3);
  int offA[BK / 16], offW[BK / 16];
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
    const int phys = (ks * 2 + hi) ^ fsw;
    offA[ks] = (wm * (BM / WM) + frow) * RB + phys * 16;
    offW[ks] = BM * RB + (wn * (BN / WN) + frow) * RB + phys * 16;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nk = K / BK;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue(s, s);

  for (int kt = 0; kt < nk; ++kt) {
    const int rem = min(NSTAGE - 2, nk - 1 - kt);  // tiles issued after kt that may stay in flight
    if (NSTAGE >= 4 && rem >= 2) wait_vmcnt<2 * LPT>();
    else if (rem >= 1) wait_vmcnt<LPT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
    const char* sbase = smem + (kt % NSTAGE) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      f16x8 a[TM], w[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) a[mi] = *reinterpret_cast<const f16x8*>(sbase + offA[ks] + mi * 32 * RB);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) w[ni] = *reinterpret_cast<const f16x8*>(sbase + offW[ks] + ni * 32 * RB);
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ni], a[mi], acc[mi][ni], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns row m, runs of 4 consecutive n.
  // Phase 1 issues every bias / residual load back-to-back (one latency, not one per group),
  // phase 2 applies bias / ReLU / residual and stores 8- or 16-byte vectors.
  const bool full_n = (n0 + BN <= e.N) && ((e.N & 3) == 0);
  if (full_n) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int nb = n0 + wn * (BN / WN) + ni * 32 + hi * 4;
      float4 bv[4];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        bv[rq] = e.bias ? *reinterpret_cast<const float4*>(e.bias + nb + rq * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 rv[TM][4];
      if (e.res) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
          const int m = m0 + wm * (BM / WM) + mi * 32 + frow;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
            rv[mi][rq] = (m < e.M) ? *reinterpret_cast<const float4*>(e.res + (size_t)m * e.ldres + nb + rq * 8)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + wm * (BM / WM) + mi * 32 + frow;
        if (m >= e.M) continue;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int n = nb + rq * 8;
          float v0 = acc[mi][ni][rq * 4 + 0] + bv[rq].x, v1 = acc[mi][ni][rq * 4 + 1] + bv[rq].y;
          float v2 = acc[mi][ni][rq * 4 + 2] + bv[rq].z, v3 = acc[mi][ni][rq * 4 + 3] + bv[rq].w;
          if (e.relu) {
            v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
          }
          if (e.res) {
            v0 += rv[mi][rq].x; v1 += rv[mi][rq].y; v2 += rv[mi][rq].z; v3 += rv[mi][rq].w;
          }
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
    return;
  }
  // ragged N tile (last column tile): per-group guards; groups are still 4-wide vectors when N%4==0
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int m = m0 + wm * (BM / WM) + mi * 32 + frow;
    if (m >= e.M) continue;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int n = n0 + wn * (BN / WN) + ni * 32 + rq * 8 + hi * 4;
        if (n >= e.N) continue;
        if (n + 3 < e.N && (e.N & 3) == 0) {
          float v0 = acc[mi][ni][rq * 4 + 0], v1 = acc[mi][ni][rq * 4 + 1];
          float v2 = acc[mi][ni][rq * 4 + 2], v3 = acc[mi][ni][rq * 4 + 3];
          if (e.bias) {
            const float4 b = *reinterpret_cast<const float4*>(e.bias + n);
            v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
          }
          if (e.relu) {
            v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
          }
          if (e.res) {
            const float4 r4 = *reinterpret_cast<const float4*>(e.res + (size_t)m * e.ldres + n);
            v0 += r4.x; v1 += r4.y; v2 += r4.z; v3 += r4.w;
          }
          if (e.C32) *reinterpret_cast<float4*>(e.C32 + (size_t)m * e.ldc32 + n) = make_float4(v0, v1, v2, v3);
          if (e.C16) {
            const __half2 h0 = __floats2half2_rn(v0, v1), h1 = __floats2half2_rn(v2, v3);
            uint2 pk;
            pk.x = *reinterpret_cast<const unsigned*>(&h0);
            pk.y = *reinterpret_cast<const unsigned*>(&h1);
            *reinterpret_cast<uint2*>(e.C16 + (size_t)m * e.ldc16 + n) = pk;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (n + i >= e.N) continue;
            float x = acc[mi][ni][rq * 4 + i] + (e.bias ? e.bias[n + i] : 0.f);
            if (e.relu) x = fmaxf(x, 0.f);
            if (e.res) x += e.res[(size_t)m * e.ldres + n + i];
            if (e.C32) e.C32[(size_t)m * e.ldc32 + n + i] = x;
            if (e.C16) e.C16[(size_t)m * e.ldc16 + n + i] = __float2half_rn(x);
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// fp16 x 3 SPLIT GEMM (LDM_PREC_SPLIT_F16): fp32-grade products on the fp16 matrix pipe.
//     x = hi + lo,   hi = fp16(x),  lo = fp16(x - hi)                            (both operands, prepared by their producers)
//     A W^T = Ahi Whi^T + Alo Whi^T + Ahi Wlo^T   (+ Alo Wlo^T ~ 2^-22 relative, dropped: below fp32 rounding)
// Same structure as gemm16_k — operands by LDS-DMA through an NSTAGE ring, one barrier per K tile, swapped-operand MFMAs so a
// lane owns one output row — with a stage holding four images (A hi | W hi | A lo | W lo) and three MFMAs per fragment
// pair on ONE accumulator (lo is unscaled: kSplitLoScale = 1, ldm_kernels.h; weights pre-scaled to magnitude ~1).  Why: gfx950's fp32 MFMA peaks at 157 TFLOP/s, its fp16 MFMA at 2 500: three
// fp16 passes have a 5.3 x higher ceiling than one fp32 pass at the same (measured: better, 7e-7 vs 9e-7) logits error, and
// 4 fragment reads feed 3 MFMAs instead of 2 feeding 1.  r03's split GEMM (gemm_f16_128x128<3>, register-staged, 221
// layouts/s) was a numerics cross-check; this one makes the split mode the fast reference-precision mode.
struct Epi16x {
  const float* bias;
  const float* res;
  float* C32;
  __half *C16, *C16lo;
  int M, N, ldres, ldc32, ldc16, relu;
  float out_scale;  // undoes the weight tensor's power-of-two pre-scale
};
constexpr float kLoScale16 = kSplitLoScale;
static_assert(kSplitLoScale == 1.0f, "gemm16x3 accumulates hi*hi, lo*hi and hi*lo into ONE accumulator: lo must be unscaled");

// Epilogue through an LDS transpose.  In the accumulator layout lane (frow, hi) owns ONE row and four 4-column runs of it, so a
// direct store is 32 rows x 8 (fp16) or 16 (fp32) bytes per wave instruction: FFN1 issued 3.0e7 such write requests per
// launch beside the 2.1e7 read requests of its operands, and these GEMMs are bound by the L2's REQUEST rate (0.66 requests
// per channel-clock in TCC_REQ, matrix pipes busy 0.28: profiles/r04_call8_*, r04_call15_*).  Here every wave parks its
// 64 x 64 tile in the (now dead) operand ring, reads it back along the rows and stores whole 128 / 256-byte row segments;
// bias / ReLU / residual are applied on the way out, the residual read is coalesced the same way.
template <int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void x3_epilogue(const f32x16 (&acc)[TM][TN], const Epi16x& e, int m0,
                                            int n0, int wm, int wn, int frow, int hi, char* smem, int wave, int lane) {
  constexpr int WR = TM * 32, WC = TN * 32, LD = WC + 4;
  float* t = reinterpret_cast<float*>(smem) + wave * (32 * LD);   // one 32-row slab of the wave's tile at a time
  __syncthreads();  // every wave is done with the operand stages
  constexpr int LPR = WC / 4;   // lanes per row
  constexpr int RPI = 64 / LPR; // rows per wave instruction
  const int lr = lane / LPR, lc = (lane % LPR) * 4;
  const int n = n0 + wn * WC + lc;
  const bool vec = (n + 3 < e.N) && ((e.N & 3) == 0);
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e.bias && n < e.N) {
    if (vec) bv = *reinterpret_cast<const float4*>(e.bias + n);
    else {
      bv.x = e.bias[n];
      if (n + 1 < e.N) bv.y = e.bias[n + 1];
      if (n + 2 < e.N) bv.z = e.bias[n + 2];
      if (n + 3 < e.N) bv.w = e.bias[n + 3];
    }
  }
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        float4 v;
        v.x = acc[mi][ni][rq * 4 + 0] * e.out_scale;
        v.y = acc[mi][ni][rq * 4 + 1] * e.out_scale;
        v.z = acc[mi][ni][rq * 4 + 2] * e.out_scale;
        v.w = acc[mi][ni][rq * 4 + 3] * e.out_scale;
        *reinterpret_cast<float4*>(t + frow * LD + ni * 32 + rq * 8 + hi * 4) = v;
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the slab is this wave's own: LDS operations of a wavefront complete in order
    __builtin_amdgcn_wave_barrier();
    if (n < e.N) {
#pragma unroll 4
      for (int it = 0; it < 32 / RPI; ++it) {
        const int r = it * RPI + lr;
        const int m = m0 + wm * WR + mi * 32 + r;
        if (m >= e.M) continue;
        const float4 tv = *reinterpret_cast<const float4*>(t + r * LD + lc);
        float v[4] = {tv.x + bv.x, tv.y + bv.y, tv.z + bv.z, tv.w + bv.w};
        if (e.relu) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if (vec) {
          if (e.res) {
            const float4 r4 = *reinterpret_cast<const float4*>(e.res + (size_t)m * e.ldres + n);
            v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
          }
          if (e.C32) *reinterpret_cast<float4*>(e.C32 + (size_t)m * e.ldc32 + n) = make_float4(v[0], v[1], v[2], v[3]);
          if (e.C16) {
            __half h[4], l[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              h[i] = __float2half_rn(v[i]);
              l[i] = __float2half_rn((v[i] - __half2float(h[i])) * kLoScale16);
            }
            *reinterpret_cast<uint2*>(e.C16 + (size_t)m * e.ldc16 + n) = *reinterpret_cast<const uint2*>(h);
            if (e.C16lo) *reinterpret_cast<uint2*>(e.C16lo + (size_t)m * e.ldc16 + n) = *reinterpret_cast<const uint2*>(l);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (n + i >= e.N) continue;
            float x = v[i];
            if (e.res) x += e.res[(size_t)m * e.ldres + n + i];
            if (e.C32) e.C32[(size_t)m * e.ldc32 + n + i] = x;
            if (e.C16) {
              const __half hh = __float2half_rn(x);
              e.C16[(size_t)m * e.ldc16 + n + i] = hh;
              if (e.C16lo) e.C16lo[(size_t)m * e.ldc16 + n + i] = __float2half_rn((x - __half2float(hh)) * kLoScale16);
            }
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // (the slab is rewritten by the next pass)
    __builtin_amdgcn_wave_barrier();
  }
}

// ABL (dev, tools/gemm_x3_probe.py): 0 the kernel; 1 no fragment reads / MFMAs (operand fills + barriers only); 2 no fills in the
// steady state (fragment reads + MFMAs + barriers on whatever the prologue left in LDS); 3 MFMAs on constant fragments (fills +
// MFMAs, no fragment reads).  Ablations produce wrong numbers by design and are reachable only through ldm_dev_bench_gemm_x3.
// HINT (dev A/B, LDM_X3_CFG 10-12): bit 0 = the ACTIVATION fills carry the non-temporal policy (aux = 2: an operand that two column
// tiles read and nobody else, straight from HBM / the Infinity Cache — FFN2's hidden activations, the out-projection's input).
template <int BM, int BN, int BK, int NSTAGE, int WM, int WN, int TAG, int ABL = 0, int HINT = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm16x3_k(const __half* __restrict__ A, const __half* __restrict__ Alo,
                                                           const __half* __restrict__ W, const __half* __restrict__ Wlo,
                                                           int lda, int ldw, int K, int tiles_n, int grp, Epi16x e) {
  constexpr int NW = WM * WN;
  constexpr int RB = BK * 2;          // bytes per tile row
  constexpr int CPR = RB / 16;        // 16-B chunks per row (4 or 8)
  constexpr int RPI = 64 / CPR;       // rows per 1-KiB DMA instruction
  constexpr int NINST = (BM + BN) / RPI;  // per half (hi / lo)
  constexpr int LPT = NINST / NW;     // DMA instructions per wave per half
  static_assert(NINST % NW == 0, "tile does not split evenly over the waves");
  constexpr int HALF_BYTES = (BM + BN) * RB;
  constexpr int STAGE_BYTES = 2 * HALF_BYTES;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // 32x32 accumulators per wave
  constexpr int JA = BM / (NW * RPI);  // instruction slots j < JA of every wave fill activation rows, the others weight rows
  static_assert(HINT == 0 || BM % (NW * RPI) == 0, "operand of a fill slot must not depend on the wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap16(blockIdx.x, gridDim.x);
  // tile order inside an XCD's contiguous range.  grp = 0: row-major (all column tiles of a row tile, then the next row
  // tile).  grp = G: column groups of G tiles, row tiles inside a group — the G x (BN x K) weight slabs of a group stay
  // in the XCD's 4-MiB L2 while the activation row tiles stream through it once per group instead of once per column tile
  int tm_i, tn_i;
  if (grp <= 0) {
    tm_i = tile / tiles_n; tn_i = tile % tiles_n;
  } else {
    const int tiles_m = gridDim.x / tiles_n;
    const int full = tiles_m * grp;                 // tiles of one full column group
    const int g = tile / full;
    const int gw = min(grp, tiles_n - g * grp);     // (the last group may be narrower)
    const int r = tile - g * full;
    tm_i = r / gw; tn_i = g * grp + r % gw;
  }
  const int m0 = tm_i * BM;
  const int n0 = tn_i * BN;

  const int lrow = lane / CPR;
  const int lchunk = lane % CPR;
  const __half *src_hi[LPT], *src_lo[LPT];
#pragma unroll
  for (int j = 0; j < LPT; ++j) {
    const int inst = wave + j * NW;
    const int R = inst * RPI + lrow;  // row in the concatenated [A rows ; W rows] image
    const bool isA = (inst * RPI) < BM;
    const int rt = isA ? R : R - BM;
    const int sw = (CPR == 8) ? ((rt >> 1) & 7) : ((rt >> 2) & 3);
    const int c = lchunk ^ sw;
    const size_t off = isA ? (size_t)(m0 + rt) * lda + c * 8 : (size_t)(n0 + rt) * ldw + c * 8;
    src_hi[j] = (isA ? A : W) + off;
    src_lo[j] = (isA ? Alo : Wlo) + off;
  }
  auto issue = [&](int kt, int stage) {
    char* sbase = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      const int inst = wave + j * NW;
      if ((HINT & 1) && j < JA) {
        __builtin_amdgcn_global_load_lds((gas_ptr)(src_hi[j] + (size_t)kt * BK), (las_ptr)(sbase + inst * 1024), 16, 0, 2);
        __builtin_amdgcn_global_load_lds((gas_ptr)(src_lo[j] + (size_t)kt * BK), (las_ptr)(sbase + HALF_BYTES + inst * 1024), 16, 0, 2);
      } else {
        __builtin_amdgcn_global_load_lds((gas_ptr)(src_hi[j] + (size_t)kt * BK), (las_ptr)(sbase + inst * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gas_ptr)(src_lo[j] + (size_t)kt * BK), (las_ptr)(sbase + HALF_BYTES + inst * 1024), 16, 0, 0);
      }
    }
  };

  const int frow = lane & 31;
  const int hi = lane >> 5;
  const int fsw = (CPR == 8) ? ((frow >> 1) & 7) : ((frow >> 2) & 3);
  int offA[BK / 16], offW[BK / 16];
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
    const int phys = (ks * 2 + hi) ^ fsw;
    offA[ks] = (wm * (BM / WM) + frow) * RB + phys * 16;
    offW[ks] = BM * RB + (wn * (BN / WN) + frow) * RB + phys * 16;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nk = K / BK;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue(s, s);

  for (int kt = 0; kt < nk; ++kt) {
    const int rem = min(NSTAGE - 2, nk - 1 - kt);  // tiles issued after kt that may stay in flight
    if (NSTAGE >= 4 && rem >= 2) wait_vmcnt<4 * LPT>();
    else if (rem >= 1) wait_vmcnt<2 * LPT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ABL != 2 && kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
    const char* sbase = smem + (kt % NSTAGE) * STAGE_BYTES;
    if constexpr (ABL == 1) continue;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      f16x8 a[TM], al[TM], w[TN], wl[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        if constexpr (ABL == 3) {
          a[mi] = al[mi] = f16x8{(_Float16)0.5f, (_Float16)0.25f, (_Float16)0.125f, (_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)0.125f, (_Float16)1.f};
          asm volatile("" : "+v"(a[mi]), "+v"(al[mi]));
          continue;
        }
        a[mi] = *reinterpret_cast<const f16x8*>(sbase + offA[ks] + mi * 32 * RB);
        al[mi] = *reinterpret_cast<const f16x8*>(sbase + HALF_BYTES + offA[ks] + mi * 32 * RB);
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        if constexpr (ABL == 3) {
          w[ni] = wl[ni] = f16x8{(_Float16)0.5f, (_Float16)0.25f, (_Float16)0.125f, (_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)0.125f, (_Float16)1.f};
          asm volatile("" : "+v"(w[ni]), "+v"(wl[ni]));
          continue;
        }
        w[ni] = *reinterpret_cast<const f16x8*>(sbase + offW[ks] + ni * 32 * RB);
        wl[ni] = *reinterpret_cast<const f16x8*>(sbase + HALF_BYTES + offW[ks] + ni * 32 * RB);
      }
      // three passes over the wave's tiles — the small cross terms first, so that they meet the running sum before the big
      // term of this k step does — instead of three back-to-back MFMAs on one accumulator: consecutive MFMAs are independent
      if constexpr (ABL != 4) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ni], a[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ni], al[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ni], a[mi], acc[mi][ni], 0, 0, 0);
      } else {  // (dev A/B: the dependent order)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ni], a[mi], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ni], al[mi], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ni], a[mi], acc[mi][ni], 0, 0, 0);
          }
      }
    }
  }

  x3_epilogue<BM, BN, WM, WN, TM, TN>(acc, e, m0, n0, wm, wn, frow, hi, smem, wave, lane);
}

// The same GEMM with the operands staged through REGISTERS (global_load_dwordx4 -> VGPRs -> ds_write_b128) instead of the LDS-DMA.
// Why: the DMA path of the chip sustains ~6.4 TB/s (MI355X_MICROARCH.md "ldsdma-fill"), and gemm16x3_k sits on it — 5-6.5 TB/s of
// operand fills, matrix pipes busy 0.28-0.36, half of the wave cycles in s_waitcnt (profiles/r04_call8_*) — while the vector
// memory path reads the L2 at several times that.  Three-stage LDS ring, ONE register stage: iteration kt computes stage kt,
// stores the registers (stage kt + 2, loaded during iteration kt - 1) and issues the loads of stage kt + 3.
// BK = 32: a 64-byte row segment per (row, stage) = four 16-byte chunks; thread t owns chunk (t & 3) of rows (t >> 2) + i * NT / 4.
template <int BM, int BN, int WM, int WN, int TAG>
__global__ __launch_bounds__(WM* WN * 64) void gemm16x3r_k(const __half* __restrict__ A, const __half* __restrict__ Alo,
                                                            const __half* __restrict__ W, const __half* __restrict__ Wlo,
                                                            int lda, int ldw, int K, int tiles_n, Epi16x e) {
  constexpr int BK = 32, NSTAGE = 3;
  constexpr int NT = WM * WN * 64;
  constexpr int RB = BK * 2;                       // 64 bytes per tile row
  constexpr int HALF_BYTES = (BM + BN) * RB;       // [A rows | W rows] of one half (hi / lo)
  constexpr int STAGE_BYTES = 2 * HALF_BYTES;
  constexpr int RPP = NT / 4;                      // rows covered by one pass of the workgroup (4 chunks per row)
  constexpr int NPA = BM / RPP, NPW = BN / RPP;    // passes over the A rows / W rows, per half
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must split evenly over the threads");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap16(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * BM;
  const int n0 = (tile % tiles_n) * BN;

  // ---- staging: global chunk (row r, chunk c) -> LDS slot r * RB + ((c ^ sw(r)) << 4), sw(r) = (r >> 2) & 3 (the read side below)
  const int srow = tid >> 2, sc = tid & 3;
  const __half* gA = A + (size_t)(m0 + srow) * lda + sc * 8;
  const __half* gAl = Alo + (size_t)(m0 + srow) * lda + sc * 8;
  const __half* gW = W + (size_t)(n0 + srow) * ldw + sc * 8;
  const __half* gWl = Wlo + (size_t)(n0 + srow) * ldw + sc * 8;
  static_assert(NPA >= 1 && NPA <= 2 && NPW >= 1 && NPW <= 2, "one or two passes per operand");
  // (named registers, not arrays: hipcc keeps lambda-captured private arrays in scratch)
  uint4 ra0, ra1, ral0, ral1, rw0, rw1, rwl0, rwl1;
  ra1 = ral1 = rw1 = rwl1 = make_uint4(0, 0, 0, 0);
  auto gload = [&](int kt) {
    const size_t ko = (size_t)kt * BK;
    ra0 = *reinterpret_cast<const uint4*>(gA + ko);
    ral0 = *reinterpret_cast<const uint4*>(gAl + ko);
    if constexpr (NPA > 1) {
      ra1 = *reinterpret_cast<const uint4*>(gA + (size_t)RPP * lda + ko);
      ral1 = *reinterpret_cast<const uint4*>(gAl + (size_t)RPP * lda + ko);
    }
    rw0 = *reinterpret_cast<const uint4*>(gW + ko);
    rwl0 = *reinterpret_cast<const uint4*>(gWl + ko);
    if constexpr (NPW > 1) {
      rw1 = *reinterpret_cast<const uint4*>(gW + (size_t)RPP * ldw + ko);
      rwl1 = *reinterpret_cast<const uint4*>(gWl + (size_t)RPP * ldw + ko);
    }
  };
  const int soff0 = srow * RB + ((sc ^ ((srow >> 2) & 3)) << 4);                   // rows srow and srow + RPP share the
  const int soff1 = (srow + RPP) * RB + ((sc ^ (((srow + RPP) >> 2) & 3)) << 4);   // swizzle when RPP % 16 == 0
  auto sstore = [&](int stage) {
    char* sb = smem + stage * STAGE_BYTES;
    *reinterpret_cast<uint4*>(sb + soff0) = ra0;
    *reinterpret_cast<uint4*>(sb + HALF_BYTES + soff0) = ral0;
    if constexpr (NPA > 1) {
      *reinterpret_cast<uint4*>(sb + soff1) = ra1;
      *reinterpret_cast<uint4*>(sb + HALF_BYTES + soff1) = ral1;
    }
    *reinterpret_cast<uint4*>(sb + BM * RB + soff0) = rw0;
    *reinterpret_cast<uint4*>(sb + HALF_BYTES + BM * RB + soff0) = rwl0;
    if constexpr (NPW > 1) {
      *reinterpret_cast<uint4*>(sb + BM * RB + soff1) = rw1;
      *reinterpret_cast<uint4*>(sb + HALF_BYTES + BM * RB + soff1) = rwl1;
    }
  };

  const int frow = lane & 31;
  const int hi = lane >> 5;
  const int fsw = (frow >> 2) & 3;
  int offA[2], offW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int phys = (ks * 2 + hi) ^ fsw;
    offA[ks] = (wm * (BM / WM) + frow) * RB + phys * 16;
    offW[ks] = BM * RB + (wn * (BN / WN) + frow) * RB + phys * 16;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nk = K / BK;
  gload(0);
  sstore(0);
  if (nk > 1) { gload(1); sstore(1); }
  if (nk > 2) gload(2);
  auto mfmas = [&](const char* sbase, int ks) {
    f16x8 a[TM], al[TM], w[TN], wl[TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      a[mi] = *reinterpret_cast<const f16x8*>(sbase + offA[ks] + mi * 32 * RB);
      al[mi] = *reinterpret_cast<const f16x8*>(sbase + HALF_BYTES + offA[ks] + mi * 32 * RB);
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      w[ni] = *reinterpret_cast<const f16x8*>(sbase + offW[ks] + ni * 32 * RB);
      wl[ni] = *reinterpret_cast<const f16x8*>(sbase + HALF_BYTES + offW[ks] + ni * 32 * RB);
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ni], a[mi], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ni], al[mi], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ni], a[mi], acc[mi][ni], 0, 0, 0);
      }
  };
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // stage kt is visible; nobody reads stage kt - 1 (= the slot of stage kt + 2) any more
    const char* sbase = smem + (kt % NSTAGE) * STAGE_BYTES;
    mfmas(sbase, 0);
    if (kt + 2 < nk) {
      sstore((kt + 2) % NSTAGE);          // (the compiler waits for the loads of stage kt + 2 here: issued one iteration ago)
      if (kt + 3 < nk) gload(kt + 3);
    }
    mfmas(sbase, 1);
  }
  x3_epilogue<BM, BN, WM, WN, TM, TN>(acc, e, m0, n0, wm, wn, frow, hi, smem, wave, lane);
}

template <int BM, int BN, int WM, int WN, int TAG>
static void launch_x3r(const GemmArgs& g, hipStream_t st) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  Epi16x e{g.bias, g.res, g.C32, g.C16, g.C16lo, g.M, g.N, g.ldres, g.ldc32, g.ldc16, g.relu, g.out_scale > 0.f ? g.out_scale : 1.0f};
  constexpr int lds = 3 * 2 * (BM + BN) * 32 * 2;
  auto kern = gemm16x3r_k<BM, BN, WM, WN, TAG>;
  allow_big_lds((const void*)kern);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WM * WN * 64), lds, st, (const __half*)g.A, (const __half*)g.Alo,
                     (const __half*)g.W, (const __half*)g.Wlo, g.lda, g.ldw, g.K, tiles_n, e);
}

template <int BM, int BN, int BK, int NSTAGE, int WM, int WN, int TAG, int ABL = 0, int HINT = 0>
static void launch_x3(const GemmArgs& g, hipStream_t st) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  Epi16x e{g.bias, g.res, g.C32, g.C16, g.C16lo, g.M, g.N, g.ldres, g.ldc32, g.ldc16, g.relu, g.out_scale > 0.f ? g.out_scale : 1.0f};
  constexpr int lds = NSTAGE * 2 * (BM + BN) * BK * 2;
  auto kern = gemm16x3_k<BM, BN, BK, NSTAGE, WM, WN, TAG, ABL, HINT>;
  allow_big_lds((const void*)kern);
  static const int grp = knob_int("LDM_X3_GRP", 0);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WM * WN * 64), lds, st, (const __half*)g.A, (const __half*)g.Alo,
                     (const __half*)g.W, (const __half*)g.Wlo, g.lda, g.ldw, g.K, tiles_n, grp, e);
}

// dev: the production tile shape with an ablation (see gemm16x3_k)
void launch_gemm16x3_abl(const GemmArgs& g, int abl, hipStream_t st) {
  const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + 255) / 256;
  Epi16x e{g.bias, g.res, g.C32, g.C16, g.C16lo, g.M, g.N, g.ldres, g.ldc32, g.ldc16, g.relu, g.out_scale > 0.f ? g.out_scale : 1.0f};
  constexpr int lds = 2 * 2 * (256 + 256) * 32 * 2;
  static const int grp = 0;
#define LDM_X3_ABL(N_)                                                                                                   \
  {                                                                                                                      \
    auto kern = gemm16x3_k<256, 256, 32, 2, 2, 4, 6, N_>;                                                                \
    allow_big_lds((const void*)kern);                                                                                    \
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, st, (const __half*)g.A, (const __half*)g.Alo,      \
                       (const __half*)g.W, (const __half*)g.Wlo, g.lda, g.ldw, g.K, tiles_n, grp, e);                    \
  }
  if (abl == 1) LDM_X3_ABL(1)
  else if (abl == 2) LDM_X3_ABL(2)
  else if (abl == 3) LDM_X3_ABL(3)
  else LDM_X3_ABL(0)
#undef LDM_X3_ABL
}

// Split-mode GEMM.  Requires: K a multiple of 32 (the host pads: Dp / Fp), A / Alo with >= ceil(M / 256) * 256 rows and
// W / Wlo with >= ceil(N / 256) * 256 rows allocated (rows past M / N feed products that are never stored).
// tag names the Linear class for rocprofv3 (0 qkv, 1 attn_out, 2 ffn1, 3 ffn2, 4 head).
void launch_gemm16x3(const GemmArgs& g, int tag, hipStream_t st) {
  // 256 x 256 tiles on 8 waves (two per SIMD; each wave 128 x 64 = 8 accumulator tiles — possible since r04's one-accumulator
  // numerics), 2-stage ring of 64-KiB stages; the vocabulary head (N = 155) keeps 256 x 128.  Same-box A/Bs
  // (profiles/r04_call6_* ... r04_call19_*): 128 x 128 606 layouts/s -> 256 x 128 706 -> coalesced epilogue 745 -> one
  // accumulator 754 -> 256 x 256 812; a 4-stage ring, 128 x 256, column-grouped tile orders, 128-byte operand rows (BK = 64),
  // operands through registers and the order of the three MFMAs change nothing or lose.
  static const int cfg = knob_int("LDM_X3_CFG", 8);  // (dev: tile-shape A/B)
  if (cfg == 0) { launch_x3<128, 128, 32, 3, 2, 2, 5>(g, st); return; }
  if (cfg == 1) { launch_x3<128, 128, 32, 4, 2, 2, 5>(g, st); return; }
  if (cfg == 2) { launch_x3<256, 128, 32, 3, 4, 2, 5>(g, st); return; }
  if (cfg == 3) { launch_x3<128, 256, 32, 3, 2, 4, 5>(g, st); return; }
  if (cfg == 5 && g.K % 64 == 0) { launch_x3<128, 128, 64, 2, 2, 2, 5>(g, st); return; }
  if (cfg == 9) { launch_x3<256, 256, 32, 2, 2, 4, 5, 4>(g, st); return; }   // the dependent MFMA order (A/B)
  if (cfg == 6) { launch_x3r<256, 128, 4, 2, 5>(g, st); return; }   // operands through registers
  if (cfg == 7) { launch_x3r<128, 128, 2, 2, 5>(g, st); return; }
  if (cfg == 10 || cfg == 11) {  // non-temporal activation fills: the two-reader operands (10), every GEMM (11)
    switch (tag) {
      case 0: if (cfg == 11) { launch_x3<256, 256, 32, 2, 2, 4, 0, 0, 1>(g, st); return; } break;
      case 1: launch_x3<256, 256, 32, 2, 2, 4, 1, 0, 1>(g, st); return;
      case 2: if (cfg == 11) { launch_x3<256, 256, 32, 2, 2, 4, 2, 0, 1>(g, st); return; } break;
      case 3: launch_x3<256, 256, 32, 2, 2, 4, 3, 0, 1>(g, st); return;
      default: break;
    }
  }
  switch (tag) {
    case 0: launch_x3<256, 256, 32, 2, 2, 4, 0>(g, st); return;
    case 1: launch_x3<256, 256, 32, 2, 2, 4, 1>(g, st); return;
    case 2: launch_x3<256, 256, 32, 2, 2, 4, 2>(g, st); return;
    case 3: launch_x3<256, 256, 32, 2, 2, 4, 3>(g, st); return;
    default: launch_x3<256, 128, 32, 3, 4, 2, 4>(g, st); return;
  }
}

template <int BM, int BN, int BK, int NSTAGE, int WM, int WN, int TAG = 0>
static void launch_cfg(const GemmArgs& g, hipStream_t st) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  Epi16 e{g.bias, g.res, g.C32, g.C16, g.M, g.N, g.ldres, g.ldc32, g.ldc16, g.relu};
  constexpr int lds = NSTAGE * (BM + BN) * BK * 2;
  auto kern = gemm16_k<BM, BN, BK, NSTAGE, WM, WN, TAG>;
  allow_big_lds((const void*)kern);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WM * WN * 64), lds, st, (const __half*)g.A,
                     (const __half*)g.W, g.lda, g.ldw, g.K, tiles_n, e);
}

// cfg ids are stable: used by the tuning hook (ldm_dev_bench_gemm) and the LDM_GEMM_CFG override
int gemm16_block_n(int cfg) {
  switch (cfg) {
    case 2: case 3: return 128;
    case 4: case 5: return 128;
    case 6: return 256;
    default: return 128;
  }
}
int gemm16_block_k(int cfg) {
  switch (cfg) {
    case 1: case 3: case 5: case 6: case 7: case 8: return 64;
    default: return 32;
  }
}

void launch_gemm16(const GemmArgs& g, int cfg, int tag, hipStream_t st) {
  if (cfg == 5) {  // production configuration: one symbol per Linear class
    switch (tag) {
      case 0: launch_cfg<128, 128, 64, 2, 2, 2, 0>(g, st); return;
      case 1: launch_cfg<128, 128, 64, 2, 2, 2, 1>(g, st); return;
      case 2: launch_cfg<128, 128, 64, 2, 2, 2, 2>(g, st); return;
      case 3: launch_cfg<128, 128, 64, 2, 2, 2, 3>(g, st); return;
      default: launch_cfg<128, 128, 64, 2, 2, 2, 4>(g, st); return;
    }
  }
  switch (cfg) {
    case 0: launch_cfg<128, 128, 32, 3, 2, 2>(g, st); break;
    case 1: launch_cfg<128, 128, 64, 3, 2, 2>(g, st); break;
    case 2: launch_cfg<128, 128, 32, 4, 2, 2>(g, st); break;
    case 3: launch_cfg<256, 128, 64, 3, 4, 2>(g, st); break;
    case 4: launch_cfg<256, 128, 32, 4, 4, 2>(g, st); break;
    case 6: launch_cfg<256, 256, 64, 2, 4, 2>(g, st); break;
    case 7: launch_cfg<256, 128, 64, 2, 4, 2>(g, st); break;
    case 8: launch_cfg<128, 256, 64, 2, 2, 4>(g, st); break;
    case 9: launch_cfg<256, 256, 32, 3, 4, 2>(g, st); break;
    case 10: launch_cfg<256, 256, 32, 4, 4, 2>(g, st); break;
    default: launch_cfg<128, 128, 32, 3, 2, 2>(g, st); break;
  }
}

}  // namespace ldm
