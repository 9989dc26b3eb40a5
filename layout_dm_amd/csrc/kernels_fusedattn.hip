// Fused QKV projection + self-attention, one workgroup per LAYOUT (125 tokens = 4 waves x 32 rows).
//
// The unfused fast path writes qkv ([M, 3*8*64] fp16, 197 MB per 512-layout chunk) and reads it back
// in the attention kernel: ~40 us of the attention launch and the whole store phase of the QKV GEMM
// are that round trip.  Here q/k/v of a layout never leave the CU:
//   * each wave keeps its 32 AdaLN-normalised token rows as MFMA operand fragments in registers
//     (normalised while loading: deferred normalisation, see kernels_rowgemm.hip);
//   * per head the six 32-row weight tiles (k0 k1 v0 v1 q0 q1) stream through the LDS-DMA ring;
//       K tile : D[i=d][j=key]   = W·X^T   -> +bias, fp16, written to Ks[key][d] in MFMA k-slot order
//       V tile : D[i=key][j=d]   = X·W^T   (operands swapped) -> V^T fragments, written to Vs[d][key-slot]
//       Q tile : D[i=d][j=query] = W·X^T   -> +bias, fp16: IS the B operand of S^T = K·Q^T (stays in VGPRs)
//   * then the attention core of kernels_attn16.hip (one 128x128 score tile, in-register softmax, P fed
//     to PV straight from the accumulators) runs on Ks/Vs (double-buffered by head parity, so the only
//     synchronisation is the ring's own per-tile barrier);
//   * output: the head-padded attention rows att[M, 8*64] fp16, exactly what attn_mfma_k produces.
// Reference semantics: torch.nn.MultiheadAttention in/out of trainer/models/transformer_utils.py:140-142,
// 197-204 with AdaLayerNorm (l.72-83) applied to the input.
#include <cstdlib>

#include "ldm_kernels.h"
#include "ldm_dma.h"
#include "ldm_pipes.h"

namespace ldm {
namespace {

constexpr int STAGE = TILE_STAGE;  // one 32-row weight tile

}  // namespace

// FUSE_OUT: additionally run the attention out-projection + residual (+ row statistics) in the same
// workgroup: att rows are written in MFMA k-slot order and read back by the SAME lanes as the B operand.
struct OutProj {
  const float* bias;   // [N]
  float* C32;          // [M, ldc] x1 = AdaLN(x) + att·Wo^T + bo
  float2* stats_out;   // [M]
  int ldc, N;
};

// phase-timing instrumentation (dev hook only, LDM_ATTN_TM=1): s_memtime sums over blocks
__device__ unsigned long long g_attn_phase[16];
#define LDM_TM_NOW() __builtin_amdgcn_s_memtime()

// V bit 0: batched prologue (row loads in 3 batches instead of 15 dependent double-buffered groups);
// V bit 1: the per-head attention outputs stay in REGISTERS until the out-projection (they already are its B
//          operand fragments): no scratch tensor, no reload phase.  Needs FUSE_OUT and H == 8.
// (The one-launch-per-layer successor of this kernel is kernels_layer.hip, the whole-stack one kernels_stack.hip.)
template <int KS, bool FUSE_OUT, bool TM = false, int V = 0>
__global__ __launch_bounds__(256, 1) void qkv_attn_k(const char* __restrict__ img, const float* __restrict__ bias,
                                                    LnLoad ln, __half* __restrict__ att, int ldo, int S, int H,
                                                    int M, float scale_log2e, OutProj op, int skew) {
  constexpr bool REG_OF = FUSE_OUT && (V & 2);
  constexpr int PF = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // smem: 2 x 32 KiB weight tiles (the ring, addressed through lds0 below), then
  char* kvbuf = smem + 2 * STAGE;          // [2 parities][Ks 16 KiB | Vs 16 KiB]
  float* sbias = reinterpret_cast<float*>(kvbuf + 4 * KV_BYTES);  // [3*H*64]
  float* sp = sbias + 3 * H * 64;          // AdaLN multiplier / shift (2 x LN_DP)
  float* sbo = sp + 2 * LN_DP;             // out-proj bias (FUSE_OUT)
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int b = blockIdx.x;

  unsigned long long t_start = 0, t_real0 = 0, s_wait = 0, s_run = 0, s_epi = 0, s_att = 0, s_w2 = 0, s_r2 = 0, s_e2 = 0;
  if constexpr (TM) {
    t_start = LDM_TM_NOW();
    t_real0 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_char_ptr)smem;
  unsigned long long t_pro = 0, t_qkv_end = 0;  // (TM; with several layers: the last layer's)
  f16x8 xf[KS];
  const int lane = tid0 & 63;
  const int tid = wave * 64 + lane;
  const int r = lane & 31, hi = lane >> 5;
  const int row_in = wave * 32 + r;          // token index inside the layout (>= S: padding)
  const bool valid = row_in < S;
  const size_t m = (size_t)b * S + (valid ? row_in : S - 1);
  const unsigned voff = lane * 16;
#pragma unroll
  for (int a = 0; a < 2; ++a) dma_lin4(voff, img + wave * 8192 + a * 4096, lds0 + wave * 8192 + a * 4096);  // tile 0
  for (int i = tid; i < 3 * H * 64; i += 256) sbias[i] = bias[i];
  // sbo = out-proj bias + AdaLN shift (the residual AdaLN(x) is recomputed in the out-proj epilogue); both
  // tables are zero beyond N / D so that padded output columns come out as exact zeros without masks
  if (FUSE_OUT)
    for (int i = tid; i < 512; i += 256) sbo[i] = i < op.N ? op.bias[i] + ln.p1[i] : 0.f;
  for (int i = tid; i < LN_DP; i += 256) {
    sp[i] = i < ln.D ? (ln.ada ? 1.0f + ln.p0[i] : ln.p0[i]) : 0.f;
    sp[LN_DP + i] = i < ln.D ? ln.p1[i] : 0.f;
  }
  __syncthreads();
  if constexpr (V & 1) {
    load_xf_ln_batched<KS, 8>(xf, ln, (int)m, hi, sp);
  } else {
    // AdaLN-on-load in groups of G k-steps, raw row loads double buffered one group ahead.  The (always zero)
    // offset is made opaque AND data-dependent on the previous group's last fragment: hipcc otherwise issues
    // all 58 row loads + 116 parameter reads up front (~700 live VGPRs -> scratch spills).
    constexpr int G = 2, NG = (KS + G - 1) / G;
    const float2 st = ln.stats[m];
    const float* xr = ln.x + m * ln.ldx + hi * 8;
    const float* mp = sp + hi * 8;
    unsigned opq = 0;
    float4 raw[2][G][2];
#pragma unroll
    for (int i = 0; i < G; ++i) {
      raw[0][i][0] = *reinterpret_cast<const float4*>(xr + i * 16);
      raw[0][i][1] = *reinterpret_cast<const float4*>(xr + i * 16 + 4);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g >= 1) asm volatile("" : "+v"(opq) : "v"(xf[g * G - 1]));
      if (g + 1 < NG) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int ks = (g + 1) * G + i;
          if (ks < KS) {
            raw[(g + 1) & 1][i][0] = *reinterpret_cast<const float4*>(xr + opq + ks * 16);
            raw[(g + 1) & 1][i][1] = *reinterpret_cast<const float4*>(xr + opq + ks * 16 + 4);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int ks = g * G + i;
        if (ks < KS) {
          const float4 a = raw[g & 1][i][0], c = raw[g & 1][i][1];
          const float4 ga = *reinterpret_cast<const float4*>(mp + opq + ks * 16);
          const float4 gb = *reinterpret_cast<const float4*>(mp + opq + ks * 16 + 4);
          const float4 sa = *reinterpret_cast<const float4*>(mp + opq + LN_DP + ks * 16);
          const float4 sb = *reinterpret_cast<const float4*>(mp + opq + LN_DP + ks * 16 + 4);
          xf[ks][0] = (_Float16)fmaf((a.x - st.x) * st.y, ga.x, sa.x);
          xf[ks][1] = (_Float16)fmaf((a.y - st.x) * st.y, ga.y, sa.y);
          xf[ks][2] = (_Float16)fmaf((a.z - st.x) * st.y, ga.z, sa.z);
          xf[ks][3] = (_Float16)fmaf((a.w - st.x) * st.y, ga.w, sa.w);
          xf[ks][4] = (_Float16)fmaf((c.x - st.x) * st.y, gb.x, sb.x);
          xf[ks][5] = (_Float16)fmaf((c.y - st.x) * st.y, gb.y, sb.y);
          xf[ks][6] = (_Float16)fmaf((c.z - st.x) * st.y, gb.z, sb.z);
          xf[ks][7] = (_Float16)fmaf((c.w - st.x) * st.y, gb.w, sb.w);
        }
      }
    }
  }
  unsigned relW[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) relW[k] = r * RKB + ((((k << 1) | hi) ^ (r & 15)) << 4);
  // weight stream: img = 32-KiB LDS images of the tiles in consumption order (ldm_api.cpp pack_attn_image):
  //   tile h*6 + j, j = k0 k1 v0 v1 q0 q1 (32 in_proj rows of head h each), then 15 out_proj tiles (K axis
  //   head-padded + k-slot ordered) and one zero tile so that the last prefetch needs no branch.
  // Row i of a tile is 1 KiB, 16-B chunk L stored at physical chunk L ^ (i & 15).
  const int n_tiles = H * 6;
  f16x8 qf[4];
  const int ksw = (r >> 1) & 7;
  const unsigned a_bias = lds0 + (unsigned)(reinterpret_cast<char*>(sbias) - smem);
  TilePipe<KS, PF> P;
  P.xf = xf;
  P.voff = voff;
  f16x8 of[32];  // (REG_OF) B-operand fragments of the out-projection: head h -> of[4h .. 4h+3]
  if constexpr (REG_OF) {
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) of[i][e] = (_Float16)0.f;
  }
  const int nskew = wave * skew;

  if constexpr (TM) t_pro = LDM_TM_NOW();
  for (int ti = 0; ti < n_tiles; ++ti) {
    const int h = ti / 6, j = ti % 6;
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0;
    if constexpr (TM) tA = LDM_TM_NOW();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int i = 0; i < nskew; ++i) asm volatile("s_nop 7");
    if constexpr (TM) {
      tB = LDM_TM_NOW();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    P.gnext = img + (size_t)(ti + 1) * STAGE + wave * 8192;  // (tile n_tiles = first out-proj tile)
    P.mnext = lds0 + ((ti + 1) & 1) * STAGE + wave * 8192;
    const unsigned sbase = lds0 + (ti & 1) * STAGE;
    char* Ks = kvbuf + (h & 1) * 2 * KV_BYTES;
    char* Vs = Ks + KV_BYTES;
#pragma unroll
    for (int k = 0; k < 8; ++k) P.aW[k] = sbase + relW[k];
    // this tile's bias: read ahead of the weight fragments (older in the LDS queue than everything the pipe
    // waits on), so the epilogue does not start with an exposed LDS round trip
    const int t = j & 1;
    float4 bpre[4];
    float bvs = 0.f;
    if (j == 2 || j == 3) {
      const unsigned ab = a_bias + (unsigned)(((2 * H + h) * 64 + t * 32 + r) * 4);
      asm volatile("ds_read_b32 %0, %1" : "=v"(bvs) : "v"(ab) : "memory");
      P.template run<true>();
    } else {
      const unsigned ab = a_bias + (unsigned)(((j < 2 ? H + h : h) * 64 + t * 32 + hi * 4) * 4);
      dsr128f<0>(bpre[0], ab);
      dsr128f<32>(bpre[1], ab);
      dsr128f<64>(bpre[2], ab);
      dsr128f<96>(bpre[3], ab);
      P.template run<false>();
    }
    const f32x16& acc = P.acc;
    if constexpr (TM) tC = LDM_TM_NOW();
    if (j < 2) {
      // K tile: lane (key = row_in, hi) holds d = 32t + 8*rq + 4*hi + i ; k-slot chunk c = 4t + 2s + hi
      {
        const f16x8 v0 = cvt8<0>(acc, bpre[0], bpre[1]);
        const f16x8 v1 = cvt8<8>(acc, bpre[2], bpre[3]);
        const int sw = (row_in >> 1) & 7;
        *reinterpret_cast<f16x8*>(Ks + row_in * 128 + (((4 * t + hi) ^ sw) << 4)) = v0;
        *reinterpret_cast<f16x8*>(Ks + row_in * 128 + (((4 * t + 2 + hi) ^ sw) << 4)) = v1;
      }
    } else if (j < 4) {
      // V tile (swapped operands): lane (d = 32t + r, hi) holds keys 32*wave + 8*rq + 4*hi + i
      const float4 b4 = make_float4(bvs, bvs, bvs, bvs);
      const int d = t * 32 + r;
      {
        const f16x8 v0 = cvt8<0>(acc, b4, b4);
        const f16x8 v1 = cvt8<8>(acc, b4, b4);
        *reinterpret_cast<f16x8*>(Vs + d * 256 + (((wave * 4 + hi) ^ (d & 15)) << 4)) = v0;
        *reinterpret_cast<f16x8*>(Vs + d * 256 + (((wave * 4 + 2 + hi) ^ (d & 15)) << 4)) = v1;
      }
    } else {
      // Q tile: stays in registers as the B operand of S^T (k-slot order = accumulator order)
      {
        const f16x8 v0 = cvt8<0>(acc, bpre[0], bpre[1]);
        const f16x8 v1 = cvt8<8>(acc, bpre[2], bpre[3]);
        if (t == 0) { qf[0] = v0; qf[1] = v1; } else { qf[2] = v0; qf[3] = v1; }
      }
    }
    if constexpr (TM) {
      tD = LDM_TM_NOW();
      s_wait += tB - tA;
      s_run += tC - tB;
      s_epi += tD - tC;
    }
    if (j != 5) continue;

    // ------------------------------------------------------------------ attention for head h
    // every wave wrote its K/V parts before the barriers of tiles j=4,5 => Ks/Vs are complete here
    f32x16 sc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[kt][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {  // ks = 2t + s ; chunk c = 4t + 2s + hi = 2*ks + hi
        const f16x8 kf =
            *reinterpret_cast<const f16x8*>(Ks + (kt * 32 + r) * 128 + (((2 * ks + hi) ^ ksw) << 4));
        sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sc[kt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = 96 + (i & 3) + 8 * (i >> 2) + 4 * hi;
      if (key >= S) sc[3][i] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sc[kt][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nmxs = -mx * scale_log2e;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sc[kt][i], scale_log2e, nmxs));
        sc[kt][i] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        f16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (_Float16)sc[kt][hf * 8 + e];
        const int c = kt * 4 + hf * 2 + hi;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int d = dt * 32 + r;
          const f16x8 vf = *reinterpret_cast<const f16x8*>(Vs + d * 256 + ((c ^ (d & 15)) << 4));
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[dt], 0, 0, 0);
        }
      }
    }
    if constexpr (REG_OF) {
      // k-slot order: fragment (dt, s) of this lane = accumulator regs 8s..8s+7 IS the out-projection's B operand
      // for k16-step 4h + 2dt + s.  A switch with constant indices per case keeps of[] in registers.
      f16x8 nf[4];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int e = 0; e < 8; ++e) nf[dt * 2 + s2][e] = (_Float16)(o[dt][s2 * 8 + e] * inv);
#define LDM_OF_CASE(HH) \
  case HH: of[4 * HH] = nf[0]; of[4 * HH + 1] = nf[1]; of[4 * HH + 2] = nf[2]; of[4 * HH + 3] = nf[3]; break;
      switch (h) {
        LDM_OF_CASE(0) LDM_OF_CASE(1) LDM_OF_CASE(2) LDM_OF_CASE(3)
        LDM_OF_CASE(4) LDM_OF_CASE(5) LDM_OF_CASE(6) LDM_OF_CASE(7)
        default: break;
      }
#undef LDM_OF_CASE
    } else if (FUSE_OUT) {
      // k-slot order: fragment (dt, s) of this lane = accumulator regs 8s..8s+7 -> 16 B at
      // [row][h*64 + dt*32 + s*16 + hi*8]; read back below by the same lane
      if (valid) {
        __half* orow = att + m * ldo + (size_t)h * 64 + hi * 8;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            f16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (_Float16)(o[dt][s2 * 8 + e] * inv);
            *reinterpret_cast<f16x8*>(orow + dt * 32 + s2 * 16) = v;
          }
      }
    } else if (valid) {
      __half* orow = att + m * ldo + (size_t)h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d = dt * 32 + rq * 8 + hi * 4;
          const __half2 h0 = __floats2half2_rn(o[dt][rq * 4 + 0] * inv, o[dt][rq * 4 + 1] * inv);
          const __half2 h1 = __floats2half2_rn(o[dt][rq * 4 + 2] * inv, o[dt][rq * 4 + 3] * inv);
          uint2 pk;
          pk.x = *reinterpret_cast<const unsigned*>(&h0);
          pk.y = *reinterpret_cast<const unsigned*>(&h1);
          *reinterpret_cast<uint2*>(orow + d) = pk;
        }
    }
    if constexpr (TM) s_att += LDM_TM_NOW() - tD;
  }
  if constexpr (TM) t_qkv_end = LDM_TM_NOW();
  if constexpr (FUSE_OUT) {
    // ------------------------------------------------------------------ out-projection phase
    if constexpr (!REG_OF) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own att stores are performed before reading back
      const __half* arow = att + m * ldo + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) of[ks] = *reinterpret_cast<const f16x8*>(arow + ks * 16);
      // consume the loads here (hipcc would otherwise put an s_waitcnt vmcnt(n) before every MFMA below)
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) asm volatile("" : "+v"(of[ks]));
    }
    const float2 rst = ln.stats[m];
    const float ra = rst.y, rb = -rst.x * rst.y;  // xn = x * ra + rb
    const float* rrow = ln.x + m * ln.ldx;
    float* crow = op.C32 + m * op.ldc + hi * 4;
    TilePipe<32, PF> Q2;
    Q2.xf = of;
    Q2.voff = voff;
    const int n_out = (op.N + 31) / 32;
    // residual row pieces (raw x) of tile ot+1 are loaded during tile ot; columns >= N are clamped to a
    // valid address (their AdaLN multiplier is 0)
    auto load_res = [&](int ot, float4(&dst)[4]) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int n = ot * 32 + g * 8 + hi * 4;
        n = n + 3 < op.N ? n : op.N - 4;
        dst[g] = *reinterpret_cast<const float4*>(rrow + n);
      }
    };
    float4 res_a[4], res_b[4];  // ping-pong (loop unrolled by two: no loop-carried register copies)
    load_res(0, res_a);
    const unsigned a_tb = lds0 + (unsigned)(reinterpret_cast<char*>(sbo) - smem) + hi * 16;
    const unsigned a_gm = lds0 + (unsigned)(reinterpret_cast<char*>(sp) - smem) + hi * 16;
    float s1 = 0.f, s2 = 0.f;
    auto out_tile = [&](int ot, float4(&rc)[4], float4(&rn)[4]) {
      const int ti = n_tiles + ot;
      unsigned long long tA = 0, tB = 0, tC = 0;
      if constexpr (TM) tA = LDM_TM_NOW();
      // oldest first: [residual of this tile x4] [DMA of this tile x8] [stores of the previous out tile x4]
      if (ot == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int i = 0; i < nskew; ++i) asm volatile("s_nop 7");
      if constexpr (TM) {
        tB = LDM_TM_NOW();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      load_res(ot + 1 < n_out ? ot + 1 : ot, rn);  // (unconditional: a branch here makes hipcc copy the fresh registers)
      __builtin_amdgcn_sched_barrier(0);
      // bias+shift and multiplier of this tile's 16 columns: read ahead of the weight fragments (older in the
      // LDS queue than everything the pipe waits on)
      float4 tbv[4], gmv[4];
      dsr128f<0>(tbv[0], a_tb + ot * 128);
      dsr128f<32>(tbv[1], a_tb + ot * 128);
      dsr128f<64>(tbv[2], a_tb + ot * 128);
      dsr128f<96>(tbv[3], a_tb + ot * 128);
      dsr128f<0>(gmv[0], a_gm + ot * 128);
      dsr128f<32>(gmv[1], a_gm + ot * 128);
      dsr128f<64>(gmv[2], a_gm + ot * 128);
      dsr128f<96>(gmv[3], a_gm + ot * 128);
      Q2.gnext = img + (size_t)(ti + 1) * STAGE + wave * 8192;  // (the tile after the last one is zero padding)
      Q2.mnext = lds0 + ((ti + 1) & 1) * STAGE + wave * 8192;
      const unsigned sbase = lds0 + (ti & 1) * STAGE;
#pragma unroll
      for (int k = 0; k < 8; ++k) Q2.aW[k] = sbase + relW[k];
      Q2.template run<false>();
      if constexpr (TM) tC = LDM_TM_NOW();
      // take delivery of the next tile's residual HERE (issued a whole MFMA run ago): hipcc's wait for it then sits
      // in front of this tile's stores, and the loop-top wait can leave those stores in flight
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(rn[g].x), "+v"(rn[g].y), "+v"(rn[g].z), "+v"(rn[g].w));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // x1 = att·Wo^T + (bo + shift) + xn * mult ; padded columns: 0 + 0 + xn * 0
        const float v0 = fmaf(fmaf(rc[g].x, ra, rb), gmv[g].x, Q2.acc[g * 4 + 0] + tbv[g].x);
        const float v1 = fmaf(fmaf(rc[g].y, ra, rb), gmv[g].y, Q2.acc[g * 4 + 1] + tbv[g].y);
        const float v2 = fmaf(fmaf(rc[g].z, ra, rb), gmv[g].z, Q2.acc[g * 4 + 2] + tbv[g].z);
        const float v3 = fmaf(fmaf(rc[g].w, ra, rb), gmv[g].w, Q2.acc[g * 4 + 3] + tbv[g].w);
        s1 += (v0 + v1) + (v2 + v3);
        s2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
        // (N % 8 == 0: a column group is entirely inside or outside for both lane halves.)  Padding rows of the
        // layout skip the store by exec mask: the instruction is still issued, so the vmcnt(4) above stays exact
        // (a fully inactive wave cannot occur: 32 rows/wave, S > 96); incomplete groups exist only in the last tile.
        if (ot * 32 + g * 8 + 8 <= op.N) {
          if (valid) *reinterpret_cast<float4*>(crow + ot * 32 + g * 8) = make_float4(v0, v1, v2, v3);
        }
      }
      if constexpr (TM) {
        s_w2 += tB - tA;
        s_r2 += tC - tB;
        s_e2 += LDM_TM_NOW() - tC;
      }
    };
    for (int ot = 0; ot < n_out; ot += 2) {
      out_tile(ot, res_a, res_b);
      if (ot + 1 < n_out) out_tile(ot + 1, res_b, res_a);
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (valid && hi == 0 && op.stats_out) {
      const float mean = s1 / (float)op.N;
      const float var = fmaxf(s2 / (float)op.N - mean * mean, 0.f);
      op.stats_out[m] = make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last (padding) tile prefetch must land before the LDS is released
  if constexpr (TM) {
    const unsigned long long t_end = LDM_TM_NOW();
    const unsigned long long t_real1 = __builtin_amdgcn_s_memrealtime();
    if (tid0 == 0) {
      atomicAdd(&g_attn_phase[0], 1ull);
      atomicAdd(&g_attn_phase[1], t_end - t_start);
      atomicAdd(&g_attn_phase[2], t_real1 - t_real0);
      atomicAdd(&g_attn_phase[3], t_pro - t_start);
      atomicAdd(&g_attn_phase[4], s_wait);
      atomicAdd(&g_attn_phase[5], s_run);
      atomicAdd(&g_attn_phase[6], s_epi);
      atomicAdd(&g_attn_phase[7], s_att);
      atomicAdd(&g_attn_phase[8], t_end - t_qkv_end);
      atomicAdd(&g_attn_phase[9], s_w2);
      atomicAdd(&g_attn_phase[10], s_r2);
      atomicAdd(&g_attn_phase[11], s_e2);
    }
  }
}

// img: tile-ordered LDS image of the layer's attention weights (see the kernel), bias: head-padded [3*H*64].
void launch_qkv_attention(const void* img, const float* bias, const LnLoad& ln, __half* att, int ldo, int B, int S,
                          int H, int dh, hipStream_t st) {
  constexpr int KS = 29;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)dh);
  const int lds = 2 * STAGE + 4 * KV_BYTES + 3 * H * 64 * 4 + 2 * LN_DP * 4 + 512 * 4;
  auto kern = qkv_attn_k<KS, false>;
  allow_big_lds((const void*)kern);
  OutProj op{};
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, st, (const char*)img, bias, ln, att, ldo, S, H, B * S, scale_log2e, op, 0);
}

// same + out-projection (tiles n_tiles.. of the image).  LDM_ATTN_V: bit 0 batched prologue, bit 1 attention outputs
// exchanged through registers (default 3); LDM_ATTN_V=0 = the r01 kernel for A/B timing.
void launch_attention_block(const void* img, const float* bias, const LnLoad& ln, __half* att, int ldo,
                            const float* b_out, float* C32, int ldc, float2* stats_out, int N,
                            int B, int S, int H, int dh, hipStream_t st) {
  constexpr int KS = 29;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)dh);
  const int lds = 2 * STAGE + 4 * KV_BYTES + 3 * H * 64 * 4 + 2 * LN_DP * 4 + 512 * 4;
  static const bool tm = getenv("LDM_ATTN_TM") && atoi(getenv("LDM_ATTN_TM")) != 0;
  static const int skew = getenv("LDM_ATTN_SKEW") ? atoi(getenv("LDM_ATTN_SKEW")) : 0;
  static const int ver_env = getenv("LDM_ATTN_V") ? atoi(getenv("LDM_ATTN_V")) : 3;
  const int ver = (H == 8) ? ver_env : (ver_env & 1);
  using K = void (*)(const char*, const float*, LnLoad, __half*, int, int, int, int, float, OutProj, int);
  K kern;
  switch (ver & 3) {
    case 0: kern = tm ? qkv_attn_k<KS, true, true, 0> : qkv_attn_k<KS, true, false, 0>; break;
    case 1: kern = tm ? qkv_attn_k<KS, true, true, 1> : qkv_attn_k<KS, true, false, 1>; break;
    default: kern = tm ? qkv_attn_k<KS, true, true, 3> : qkv_attn_k<KS, true, false, 3>; break;
  }
  allow_big_lds((const void*)kern);
  OutProj op{b_out, C32, stats_out, ldc, N};
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, st, (const char*)img, bias, ln, att, ldo, S, H, B * S, scale_log2e, op,
                     skew);
}

void attn_phase_read(unsigned long long* out16) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_attn_phase), 16 * sizeof(unsigned long long));
  unsigned long long z[16] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_phase), z, sizeof(z));
}

}  // namespace ldm
