// Fused plain-fp16 FFN of the HYBRID numerics mode (LDM_PREC_HYBRID_F16), one launch per block:
//
//   out[M, 464] = x + b2 + W2 · relu(W1 · LN2(x) + b1)          x = the block's attention output (fp32 rows), out = the next block's input
//
// (transformer_utils.py:165-210 _ff_block; Block.forward's second residual.)  The hybrid mode rounds the FFN's operands once — LayerNorm-2 output,
// hidden activations and weights are plain fp16, products accumulate in fp32 — which is exactly what the fast mode's stack kernel does for ITS FFN, so
// this kernel is that phase of kernels_stack.hip lifted out as a row kernel: a workgroup owns 128 rows, their residual sum lives in 15 accumulator
// tiles (AGPRs) from the first load to the last store, the normalised rows are 29 fp16 fragments in registers, and the weights stream through the LDS
// as the fast mode's chunk image (ldm_pack::pack_ffn_image_pipelined: stage i = W1 tile i | W2 slab i - 1, 64 KiB, linear LDS-DMA into a two-stage
// ring) under ldm_pipes.h FfnStream — one continuous LDS-read / MFMA pipeline, 59 MFMAs per 32 hidden units, the bias / ReLU / cast of chunk i in the
// MFMA shadows of chunk i - 1's second GEMM.  The hidden activations never exist in memory: against the two launches it replaces (linear1 writing
// plain-fp16 panels; linear2 as the GEMM prologue of the next launch) that is 119 MB less written and 119 MB less read per block and 256 layouts.
// gfx950 only; geometry: D == 464, F % 32 == 0, F <= 2048.
#include "ldm_dma.h"
#include "ldm_kernels.h"
#include "ldm_pipes.h"

namespace ldm {

namespace {
constexpr int FR_KS = 29, FR_NT2 = 15, FR_NGV = 58;
constexpr int FR_PAR_OFF = 2 * FFN_STAGE;                 // norm2 gamma [512] | beta [512]
constexpr int FR_B2_OFF = FR_PAR_OFF + 2 * LN_DP * 4;     // linear2 bias [512]
constexpr int FR_B1_OFF = FR_B2_OFF + 512 * 4;            // linear1 bias [2048]
constexpr int FR_LDS = FR_B1_OFF + 2048 * 4;              // 145 408 B
static_assert(FR_LDS <= 160 * 1024, "LDS budget");
}  // namespace

__global__ __launch_bounds__(256, 1) void ffn16_rows_k(FfnRowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, hi = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(lds_char_ptr)smem;
  const unsigned voff = (unsigned)lane * 16;
  float* sp2 = reinterpret_cast<float*>(smem + FR_PAR_OFF);
  float* sb2 = reinterpret_cast<float*>(smem + FR_B2_OFF);
  float* sb1 = reinterpret_cast<float*>(smem + FR_B1_OFF);

  // ---- chunk 0 -> stage 0 (this wave's 16 KiB); lands while the rows are read and normalised
  {
    const char* g0 = a.img + wave * 16384;
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_lin4(voff, g0 + k * 4096, lds0 + wave * 16384 + k * 4096);
  }
  // ---- parameter tables -> LDS (zero beyond D / F: padded columns and hidden units come out as exact zeros)
  for (int i = tid; i < LN_DP; i += 256) {
    const bool in = i < a.D;
    sp2[i] = in ? a.gamma[i] : 0.f;
    sp2[LN_DP + i] = in ? a.beta[i] : 0.f;
    sb2[i] = in ? a.b2[i] : 0.f;
  }
  for (int i = tid; i < 2048; i += 256) sb1[i] = i < a.F ? a.b1[i] : 0.f;

  // ---- the rows, raw, in accumulator layout: lane (row, hi) owns columns 8 g + 4 hi .. + 3 of every 8-column group g
  const int row = blockIdx.x * 128 + wave * 32 + r;
  const int rrow = row < a.M ? row : a.M - 1;
  float4 v[FR_NGV];
  {
    const float* x = a.x + (size_t)rrow * a.D + hi * 4;
#pragma unroll
    for (int g = 0; g < FR_NGV; ++g) v[g] = *reinterpret_cast<const float4*>(x + g * 8);
  }
  // two-pass statistics over the row (this lane's half + lane ^ 32), eps 1e-5 — as kernels_lngemm.hip / ln_rows
  float s1 = 0.f;
#pragma unroll
  for (int g = 0; g < FR_NGV; ++g) s1 += (v[g].x + v[g].y) + (v[g].z + v[g].w);
  s1 += __shfl_xor(s1, 32, 64);
  const float inv_d = 1.0f / (float)a.D;
  const float mean = s1 * inv_d;
  float s2 = 0.f;
#pragma unroll
  for (int g = 0; g < FR_NGV; ++g) {
    const float dx = v[g].x - mean, dy = v[g].y - mean, dz = v[g].z - mean, dw = v[g].w - mean;
    s2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  s2 += __shfl_xor(s2, 32, 64);
  const float rstd = 1.0f / sqrtf(s2 * inv_d + 1e-5f);
  __syncthreads();  // parameter tables visible

  // ---- normalised fp16 fragments in k-slot order (groups 2 ks, 2 ks + 1 of the accumulator layout ARE fragment ks) and the second GEMM's seed
  // acc = x + b2, tile by tile into the AGPRs (whole-tuple moves: with element-wise accumulator updates hipcc split live ranges through scratch)
  f16x8 xf[FR_KS];
  f32x16 acc[FR_NT2];
  {
    const float* gp = sp2 + hi * 4;
    const float* bp = sb2 + hi * 4;
#pragma unroll
    for (int t = 0; t < FR_NT2; ++t) {
      f32x16 tile;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int gg = t * 4 + g;
        if (gg < FR_NGV) {
          const int ks = gg >> 1, e0 = (gg & 1) * 4;
          const float4 ga = *reinterpret_cast<const float4*>(gp + gg * 8);
          const float4 be = *reinterpret_cast<const float4*>(gp + LN_DP + gg * 8);
          const float4 bb = *reinterpret_cast<const float4*>(bp + gg * 8);
          xf[ks][e0 + 0] = (_Float16)((v[gg].x - mean) * rstd * ga.x + be.x);
          xf[ks][e0 + 1] = (_Float16)((v[gg].y - mean) * rstd * ga.y + be.y);
          xf[ks][e0 + 2] = (_Float16)((v[gg].z - mean) * rstd * ga.z + be.z);
          xf[ks][e0 + 3] = (_Float16)((v[gg].w - mean) * rstd * ga.w + be.w);
          tile[g * 4 + 0] = v[gg].x + bb.x;
          tile[g * 4 + 1] = v[gg].y + bb.y;
          tile[g * 4 + 2] = v[gg].z + bb.z;
          tile[g * 4 + 3] = v[gg].w + bb.w;
          if (gg & 1) asm volatile("" : "+v"(xf[ks]));
        } else {
          tile[g * 4 + 0] = tile[g * 4 + 1] = tile[g * 4 + 2] = tile[g * 4 + 3] = 0.f;
        }
      }
      asm volatile("" : "+a"(tile));
      acc[t] = tile;
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- the chunk stream (ldm_pipes.h FfnStream, software-pipelined form)
  {
    unsigned relW1[8], relW2[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) relW1[k] = r * RKB + ((((k << 1) | hi) ^ (r & 15)) << 4);
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) relW2[sx] = r * 64 + (((2 * sx + hi) ^ ((r >> 2) & 3)) << 4);
    const unsigned relB = lds0 + FR_B1_OFF + hi * 16;
    FfnStream<FR_KS, FR_NT2, 2, false, 6, true> F;
    F.xf = xf;
    F.acc = acc;
    F.voff = voff;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // chunk 0 (own pieces), then everybody's
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) F.aW1[k] = lds0 + relW1[k];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) F.aW2[sx] = lds0 + relW2[sx];
    F.ab_next = relB;
    // every fragment back in its registers, hipcc's scoreboard drained
#pragma unroll
    for (int k = 0; k < FR_KS; ++k) asm volatile("" : "+v"(xf[k]));
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_sched_barrier(0);
    F.read_bias();
    F.template prologue<0>();
    {
      const f16x8 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      F.pf[0] = F.pf[1] = F.pfn[0] = F.pfn[1] = z;  // iteration 0's second GEMM multiplies zero fragments (and a zero W2 half)
    }
    for (int c = 0; c <= a.n_chunks; ++c) {   // iteration c = GEMM1 of chunk c + GEMM2 of chunk c - 1 on stage c
      F.gnext = a.img + (size_t)(c == a.n_chunks ? 0 : c + 1) * FFN_STAGE + wave * 16384;
      F.mnext = lds0 + ((c + 1) & 1) * FFN_STAGE + wave * 16384;
      F.ab_next = relB + (c + 1 >= a.n_chunks ? 0 : c + 1) * 128;
      F.template step<0, true>();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- out rows = acc (x + b2 + W2 relu(...)): 16-byte pieces of the lane's row, like the AdaLN rows kernels_lngemm.hip writes
  if (row < a.M) {
    float* o = a.out + (size_t)row * a.D + hi * 4;
#pragma unroll
    for (int t = 0; t < FR_NT2; ++t) {
      const f32x16 tile = acc[t];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int gg = t * 4 + g;
        if (gg < FR_NGV) *reinterpret_cast<float4*>(o + gg * 8) = make_float4(tile[g * 4 + 0], tile[g * 4 + 1], tile[g * 4 + 2], tile[g * 4 + 3]);
      }
    }
  }
}

int launch_ffn16_rows(const FfnRowsArgs& a, hipStream_t st) {
  if (a.D != 464 || a.F < 32 || (a.F & 31) || a.F > 2048 || a.n_chunks != a.F / 32 || a.M < 1 || !a.x || !a.out || !a.img || !a.gamma || !a.beta ||
      !a.b1 || !a.b2)
    return -1;
  allow_big_lds((const void*)ffn16_rows_k);
  hipLaunchKernelGGL(ffn16_rows_k, dim3((a.M + 127) / 128), dim3(256), FR_LDS, st, a);
  return 0;
}

}  // namespace ldm
