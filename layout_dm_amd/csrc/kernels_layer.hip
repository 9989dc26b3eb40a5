// One transformer layer per launch, one workgroup per LAYOUT — the stream version of the fused layer kernel
// (kernels_fusedattn.hip qkv_attn_k<..., 11>, same data layout, same weight images, same arithmetic up to the
// placement of the V bias):
//
//   prologue   AdaLN-on-load of the wave's 32 rows into 29 fp16 MFMA fragments (registers)
//   per head   HeadStream: k0 k1 v0 v1 q0 q1 as ONE continuous 174-item LDS-read / MFMA pipeline (no per-tile queue
//              restart, tile epilogues in the next tile's MFMA shadow, K/Q bias through the MFMA C operand), then the
//              single-tile attention core (in-register softmax, P fed to PV from the accumulators, V bias added to the
//              normalised output: softmax rows sum to 1); output fragments parked in AGPRs
//   out-proj   residual seed AdaLN(x) + b_out in 15 persistent accumulator tiles, then SlabStream: the 16 K-slabs as ONE
//              continuous 480-item pipeline
//   LN2 + FFN  statistics / fragments / GEMM2 seed from the accumulators, FfnStream chunk loop
//   epilogue   row statistics + stores of x2 (in place)
//
// Reference semantics: Block.forward, trainer/models/transformer_utils.py:165-210 (AdaLayerNorm l.72-83,
// nn.MultiheadAttention l.140-142, FFN l.179, 208-209).
#include <cstdlib>

#include "ldm_kernels.h"
#include "ldm_dma.h"
#include "ldm_pipes.h"

namespace ldm {

struct LayerArgs {
  const char* img;      // pack_attn_slab_image: 6H in_proj tiles, 16 out-proj K-slabs, 1 zero stage (32 KiB each)
  const float* bias;    // head-padded in_proj bias [3*H*64]
  LnLoad ln;            // AdaLN of the layer input: x rows, (mean, rstd), scale, shift
  const float* b_out;   // [N] out_proj bias + W_out b_v (V bias folded on the host, ldm_api.cpp build_fast_weights)
  const char* ffn_img;  // pack_ffn_image (W1 K axis k-slot ordered), 64 KiB per 32-wide hidden chunk
  const float *b1, *b2, *g2, *be2;
  float* out;           // [M, ldo] x2 (may alias ln.x)
  float2* stats_out;    // [M]
  int ldo, N, S, H, n_chunks;
  float scale_log2e;
};

__device__ unsigned long long g_layer_phase[16];

__device__ __forceinline__ int hw_lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <bool TM, int DBG = 0>
__global__ __launch_bounds__(256, 1) void layer_stream_k(LayerArgs a) {
  constexpr int KS = 29, STAGE = TILE_STAGE, NT2 = 15, NGV = 58;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kvbuf = smem + 3 * STAGE;          // Ks 16 KiB | Vs 16 KiB behind the 3-stage weight ring
  float* sbias = reinterpret_cast<float*>(kvbuf + 2 * KV_BYTES);  // [3*H*64]
  float* sp = sbias + 3 * a.H * 64;        // AdaLN multiplier / shift (2 x LN_DP)
  float* sbo = sp + 2 * LN_DP;             // out-proj bias + AdaLN shift [512]
  float* sb1 = sbo + 512;                  // linear1 bias [n_chunks*32]
  float* sp2 = sb1 + a.n_chunks * 32;      // norm2 gamma | beta (2 x LN_DP)
  float* sb2 = sp2 + 2 * LN_DP;            // linear2 bias [512], zero beyond N
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int b = blockIdx.x, S = a.S, H = a.H;
  const unsigned lds0 = (unsigned)(size_t)(lds_char_ptr)smem;
  const unsigned voff = lane * 16;

  unsigned long long t_start = 0, t_real0 = 0, t_pro = 0, s_stream = 0, s_core = 0, t_heads = 0, s_hsync = 0, s_ssync = 0, s_fsync = 0;
  if constexpr (TM) {
    t_start = __builtin_amdgcn_s_memtime();
    t_real0 = __builtin_amdgcn_s_memrealtime();
  }
  f16x8 xf[KS];
  {
    const int r = lane & 31, hi = lane >> 5;
    const int row_in = wave * 32 + r;
    const size_t m = (size_t)b * S + (row_in < S ? row_in : S - 1);
#pragma unroll
    for (int k = 0; k < 2; ++k) dma_lin4(voff, a.img + wave * 8192 + k * 4096, lds0 + wave * 8192 + k * 4096);  // tile 0
#pragma unroll
    for (int k = 0; k < 2; ++k)  // tile 1 -> stage 1
      dma_lin4(voff, a.img + STAGE + wave * 8192 + k * 4096, lds0 + STAGE + wave * 8192 + k * 4096);
    for (int i = tid; i < 3 * H * 64; i += 256) sbias[i] = a.bias[i];
    // sbo = out-proj bias + AdaLN shift (the residual AdaLN(x) is recomputed at the residual seed); every table is zero
    // beyond N / D so that padded output columns come out as exact zeros without masks
    for (int i = tid; i < 512; i += 256) sbo[i] = i < a.N ? a.b_out[i] + a.ln.p1[i] : 0.f;
    for (int i = tid; i < LN_DP; i += 256) {
      sp[i] = i < a.ln.D ? 1.0f + a.ln.p0[i] : 0.f;
      sp[LN_DP + i] = i < a.ln.D ? a.ln.p1[i] : 0.f;
      sp2[i] = i < a.N ? a.g2[i] : 0.f;
      sp2[LN_DP + i] = i < a.N ? a.be2[i] : 0.f;
      sb2[i] = i < a.N ? a.b2[i] : 0.f;
    }
    for (int i = tid; i < a.n_chunks * 32; i += 256) sb1[i] = a.b1[i];
    __syncthreads();
    // (r02 negative result: issuing 40 of the 58 row loads before the table set-up and the rest two batches deep made
    //  this phase 12 % LONGER — 32.2k vs 28.8k cycles: it is paced by the all-CU burst on HBM, not by the number of
    //  exposed round trips: profiles/r02_call21_*)
    load_xf_ln_batched<KS, 8>(xf, a.ln, (int)m, hi, sp);
  }
  if constexpr (TM) t_pro = __builtin_amdgcn_s_memtime();

  f16x8 of[32];  // B-operand fragments of the out-projection: head h -> of[4h .. 4h+3]
#pragma unroll
  for (int i = 0; i < 32; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) of[i][e] = (_Float16)0.f;
  {
    // ------------------------------------------------------------------ QKV + attention, head by head
    const int r = lane & 31, hi = lane >> 5;
    const int row_in = wave * 32 + r;
    f16x8 qf[4];
    HeadStream<TM> HS;
    HS.xf = xf;
    HS.qf = qf;
    HS.voff = voff;
    HS.lds_w = lds0 + wave * 8192;
    HS.H = H;
    HS.a_bias = lds0 + (unsigned)(reinterpret_cast<char*>(sbias) - smem) + hi * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      HS.aW[k] = lds0 + r * RKB + ((((k << 1) | hi) ^ (r & 15)) << 4);
      HS.aW2[k] = HS.aW[k] + 2 * STAGE;
    }
    const unsigned kv0 = lds0 + 3 * STAGE;  // Ks 16 KiB | Vs 16 KiB (single-buffered: see the header comment)
    {
      const int sw = (row_in >> 1) & 7;
      HS.aK[0] = kv0 + row_in * 128 + ((hi ^ sw) << 4);
      HS.aK[1] = kv0 + row_in * 128 + (((2 + hi) ^ sw) << 4);
      HS.aV[0] = kv0 + KV_BYTES + r * 256 + (((wave * 4 + hi) ^ (r & 15)) << 4);
      HS.aV[1] = kv0 + KV_BYTES + r * 256 + (((wave * 4 + 2 + hi) ^ (r & 15)) << 4);
    }
    AttnCoreV AC;  // (the all-VGPR core of the stack kernel: one implementation of the attention core)
    AC.qf = qf;
    AC.aKr = kv0 + r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);
    AC.aVr = kv0 + KV_BYTES + r * 256 + ((hi ^ (r & 15)) << 4);
    AC.scale_log2e = a.scale_log2e;
    AC.S = S;
    AC.hi = hi;
    // tiles 0 and 1 have landed (own pieces, then everybody's)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int h = 0; h < H; ++h) {
      unsigned long long tA = 0, tB = 0;
      if constexpr (TM) tA = __builtin_amdgcn_s_memtime();
      HS.h = h;
      HS.gimg = a.img + (size_t)h * 6 * STAGE + wave * 8192;
      HS.run();
      if constexpr (TM) tB = __builtin_amdgcn_s_memtime();
      // ---------------------------------------------------------------- attention core of head h
      // (every wave wrote its K / V parts before the barriers of tiles q0, q1 => Ks / Vs are complete here; they are
      //  not overwritten before every wave is past this core: the next K write sits behind the next head's first
      //  barrier)
      f16x8 nf[4];
      if constexpr (DBG & 1) {
        // (A/B aid, LDM_LAYER_DBG=1) the compiler-scheduled core of the first stream version
        const char* Ks = kvbuf;
        const char* Vs = Ks + KV_BYTES;
        const int ksw = (r >> 1) & 7;
        f32x16 sc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
          for (int i = 0; i < 16; ++i) sc[kt][i] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const f16x8 kf = *reinterpret_cast<const f16x8*>(Ks + (kt * 32 + r) * 128 + (((2 * ks + hi) ^ ksw) << 4));
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
        const float nmxs = -mx * a.scale_log2e;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sc[kt][i], a.scale_log2e, nmxs));
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
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) nf[dt * 2 + (i >> 3)][i & 7] = (_Float16)(o[dt][i] * inv);
      } else {
        AC.run(nf);
      }
      // park them in AGPRs (the arch VGPRs belong to the activation fragments of the streams), in place: see park_agpr4
#define LDM_OF_CASE(HH) \
  case HH: park_agpr4(of[4 * HH], nf[0]); park_agpr4(of[4 * HH + 1], nf[1]); park_agpr4(of[4 * HH + 2], nf[2]); \
           park_agpr4(of[4 * HH + 3], nf[3]); break;
      switch (h) {
        LDM_OF_CASE(0) LDM_OF_CASE(1) LDM_OF_CASE(2) LDM_OF_CASE(3)
        LDM_OF_CASE(4) LDM_OF_CASE(5) LDM_OF_CASE(6) LDM_OF_CASE(7)
        default: break;
      }
#undef LDM_OF_CASE
      if constexpr (TM) {
        s_hsync = HS.t_sync;
        s_stream += tB - tA;
        s_core += __builtin_amdgcn_s_memtime() - tB;
      }
    }
  }
  if constexpr (TM) t_heads = __builtin_amdgcn_s_memtime();

  // ==================================================================== out-projection (K slabs) -> LN2 -> FFN
  // lane coordinates re-derived from the hardware lane id behind an opaque asm: hipcc otherwise hoists the address
  // arithmetic of the phases below above the head loop, where every register is taken
  const int lane1 = hw_lane_id();
  const int r1 = lane1 & 31, hi1 = lane1 >> 5;
  const int row1 = wave * 32 + r1;
  const size_t m1 = (size_t)b * S + (row1 < S ? row1 : S - 1);
  f32x16 acc[NT2];
#pragma unroll
  for (int t = 0; t < NT2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  {
    // residual seed in accumulator layout: AdaLN(x)[row][cols] + (b_out + shift)[cols]; lane (row, hi) owns columns
    // 8g + 4hi .. +3 of every 8-column group g
    constexpr int GB = 20;
    const float2 rst = a.ln.stats[m1];
    const float ra = rst.y, rb = -rst.x * rst.y;  // xn = x * ra + rb
    const float* rrow = a.ln.x + m1 * a.ln.ldx + hi1 * 4;
    const float* gmp = sp + hi1 * 4;
    const float* tbp = sbo + hi1 * 4;
#pragma unroll
    for (int g0 = 0; g0 < NGV; g0 += GB) {
      float4 raw[GB];
#pragma unroll
      for (int i = 0; i < GB; ++i)
        if (g0 + i < NGV) raw[i] = *reinterpret_cast<const float4*>(rrow + (g0 + i) * 8);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        const int gg = g0 + i;
        if (gg < NGV) {
          const float4 x = raw[i];
          const float4 gm = *reinterpret_cast<const float4*>(gmp + gg * 8);
          const float4 tb = *reinterpret_cast<const float4*>(tbp + gg * 8);
          const int t = gg >> 2, q0 = (gg & 3) * 4;
          acc[t][q0 + 0] = to_agpr(fmaf(fmaf(x.x, ra, rb), gm.x, tb.x));
          acc[t][q0 + 1] = to_agpr(fmaf(fmaf(x.y, ra, rb), gm.y, tb.y));
          acc[t][q0 + 2] = to_agpr(fmaf(fmaf(x.z, ra, rb), gm.z, tb.z));
          acc[t][q0 + 3] = to_agpr(fmaf(fmaf(x.w, ra, rb), gm.w, tb.w));
          if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  {
    // 16 K slabs (k chunk c = head c/2, d-half c%2; B operands of[2c], of[2c+1]); slab 0 was prefetched by the last
    // q tile (image stage 6H, certified by that tile's barrier)
    SlabStream<NT2, TM> SS;
    SS.acc = acc;
    SS.of = of;
    SS.voff = voff;
    SS.lds_w = lds0 + wave * 8192;
    SS.gimg = a.img + (size_t)(6 * H) * STAGE + wave * 8192;
    SS.aS[0] = lds0 + r1 * 64 + (((0 + hi1) ^ ((r1 >> 2) & 3)) << 4);
    SS.aS[1] = lds0 + r1 * 64 + (((2 + hi1) ^ ((r1 >> 2) & 3)) << 4);
    SS.aS2[0] = SS.aS[0] + 2 * STAGE;
    SS.aS2[1] = SS.aS[1] + 2 * STAGE;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SS.template run<0>();
    if constexpr (TM) s_ssync = SS.t_sync;
  }
  unsigned long long t_slab = 0, t_ln2 = 0, t_ffn = 0;
  if constexpr (TM) t_slab = __builtin_amdgcn_s_memtime();
  // acc = x1 (rows of this layout).  Everybody is done with the attention ring / K,V buffers after this barrier:
  // the FFN ring (2 x 64 KiB at LDS 0) takes their place.
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {  // FFN chunk 0 -> stage 0 (this wave's 16 KiB); lands while LN2 runs
    const char* g0 = a.ffn_img + wave * 16384;
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_lin4(voff, g0 + k * 4096, lds0 + wave * 16384 + k * 4096);
  }
  const int lane2 = hw_lane_id();
  const int r2 = lane2 & 31, hi2 = lane2 >> 5;
  f16x8 xf2[KS];
  {
    // LN2 statistics of the row (this lane's half + lane^32), normalised fp16 fragments in k-slot order (groups
    // 2ks, 2ks+1 of the accumulator layout ARE fragment ks), GEMM2 seed acc = x1 + b2
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int gg = 0; gg < NGV; ++gg) {
      const int t = gg >> 2, q0 = (gg & 3) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = acc[t][q0 + i];
        s1 += v;
        s2 += v * v;
      }
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    constexpr float kInvN = 1.0f / 464.0f;  // N = 464 (launcher)
    const float mean = s1 * kInvN;
    const float rstd = 1.0f / sqrtf(fmaxf(s2 * kInvN - mean * mean, 0.f) + 1e-5f);
    const float* gp = sp2 + hi2 * 4;
    const float* bp = sb2 + hi2 * 4;
#pragma unroll
    for (int gg = 0; gg < NGV; ++gg) {
      const int t = gg >> 2, q0 = (gg & 3) * 4, ks = gg >> 1, e0 = (gg & 1) * 4;
      const float4 ga = *reinterpret_cast<const float4*>(gp + gg * 8);
      const float4 be = *reinterpret_cast<const float4*>(gp + LN_DP + gg * 8);
      const float4 bb = *reinterpret_cast<const float4*>(bp + gg * 8);
      const float v0 = acc[t][q0 + 0], v1 = acc[t][q0 + 1], v2 = acc[t][q0 + 2], v3 = acc[t][q0 + 3];
      xf2[ks][e0 + 0] = (_Float16)fmaf((v0 - mean) * rstd, ga.x, be.x);
      xf2[ks][e0 + 1] = (_Float16)fmaf((v1 - mean) * rstd, ga.y, be.y);
      xf2[ks][e0 + 2] = (_Float16)fmaf((v2 - mean) * rstd, ga.z, be.z);
      xf2[ks][e0 + 3] = (_Float16)fmaf((v3 - mean) * rstd, ga.w, be.w);
      acc[t][q0 + 0] = to_agpr(v0 + bb.x);
      acc[t][q0 + 1] = to_agpr(v1 + bb.y);
      acc[t][q0 + 2] = to_agpr(v2 + bb.z);
      acc[t][q0 + 3] = to_agpr(v3 + bb.w);
      if (gg & 1) asm volatile("" : "+v"(xf2[ks]));
      if ((gg & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (TM) t_ln2 = __builtin_amdgcn_s_memtime();
  {
    // ---- FFN chunk loop: one continuous LDS-read / MFMA pipeline (ldm_pipes.h FfnStream)
    unsigned relW1[8], relW2[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) relW1[k] = r2 * RKB + ((((k << 1) | hi2) ^ (r2 & 15)) << 4);
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) relW2[sx] = r2 * 64 + (((2 * sx + hi2) ^ ((r2 >> 2) & 3)) << 4);
    const unsigned relB = lds0 + (unsigned)(reinterpret_cast<char*>(sb1) - smem) + hi2 * 16;
    FfnStream<KS, NT2, 2, TM> F;
    F.xf = xf2;
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
    F.read_bias();
    F.template prologue<0>();
    for (int c = 0; c < a.n_chunks; ++c) {
      F.gnext = a.ffn_img + (size_t)(c + 1 == a.n_chunks ? 0 : c + 1) * FFN_STAGE + wave * 16384;
      F.mnext = lds0 + ((c + 1) & 1) * FFN_STAGE + wave * 16384;
      F.ab_next = relB + (c + 1 == a.n_chunks ? 0 : c + 1) * 128;
      F.template step<0, true>();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TM) s_fsync = F.t_sync;
  }
  if constexpr (TM) t_ffn = __builtin_amdgcn_s_memtime();
  {
    // ---- x2 = acc: row statistics + stores (only the rows of this layout: padding rows of the last wave belong to
    // the next layout)
    const int lane3 = hw_lane_id();
    const int r3 = lane3 & 31, hie = lane3 >> 5;
    const int row3 = wave * 32 + r3;
    const bool valid3 = row3 < S;
    const size_t me = (size_t)b * S + (valid3 ? row3 : S - 1);
    float* orow = a.out + me * a.ldo + hie * 4;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int gg = 0; gg < NGV; ++gg) {
      const int t = gg >> 2, q0 = (gg & 3) * 4;
      const float v0 = acc[t][q0 + 0], v1 = acc[t][q0 + 1], v2 = acc[t][q0 + 2], v3 = acc[t][q0 + 3];
      s1 += (v0 + v1) + (v2 + v3);
      s2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      if (valid3) *reinterpret_cast<float4*>(orow + gg * 8) = make_float4(v0, v1, v2, v3);
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    constexpr float kInvN3 = 1.0f / 464.0f;
    const float mean3 = s1 * kInvN3;
    const float rstd3 = 1.0f / sqrtf(fmaxf(s2 * kInvN3 - mean3 * mean3, 0.f) + 1e-5f);
    if (valid3 && hie == 0 && a.stats_out) a.stats_out[me] = make_float2(mean3, rstd3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last FFN prefetch must land before the LDS is released
  if constexpr (TM) {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    const unsigned long long t_real1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      atomicAdd(&g_layer_phase[0], 1ull);
      atomicAdd(&g_layer_phase[1], t_end - t_start);
      atomicAdd(&g_layer_phase[2], t_real1 - t_real0);
      atomicAdd(&g_layer_phase[3], t_pro - t_start);
      atomicAdd(&g_layer_phase[4], s_stream);
      atomicAdd(&g_layer_phase[5], s_core);
      atomicAdd(&g_layer_phase[6], t_slab - t_heads);
      atomicAdd(&g_layer_phase[7], t_ln2 - t_slab);
      atomicAdd(&g_layer_phase[8], t_ffn - t_ln2);
      atomicAdd(&g_layer_phase[9], t_end - t_ffn);
      atomicAdd(&g_layer_phase[10], s_hsync);
      atomicAdd(&g_layer_phase[11], s_ssync);
      atomicAdd(&g_layer_phase[12], s_fsync);
    }
  }
}

// One transformer layer per launch: x <- x2 in place.  img: pack_attn_slab_image, ffn_img: pack_ffn_image with W1's K
// axis in k-slot order.  ln: AdaLN of the layer input (x rows, stats = stats_io), N = d_model = 464, 8 heads.
int layer_stream_debug() {
  static const int dbg = getenv("LDM_LAYER_DBG") ? atoi(getenv("LDM_LAYER_DBG")) : 0;
  return dbg;
}

void launch_layer_stream(const void* img, const float* bias, const LnLoad& ln, const float* b_out, const void* ffn_img,
                         const float* b1, const float* b2, const float* g2, const float* be2, int F, float* x, int ldx,
                         float2* stats_io, int N, int B, int S, int H, int dh, hipStream_t st) {
  const int lds = 3 * TILE_STAGE + 2 * KV_BYTES + (3 * H * 64 + 2 * LN_DP + 512 + F + 2 * LN_DP + 512) * 4;
  static const bool tm = getenv("LDM_ATTN_TM") && atoi(getenv("LDM_ATTN_TM")) != 0;
  auto kern = tm ? layer_stream_k<true> : layer_stream_k<false>;
  switch (layer_stream_debug()) {
    case 1: kern = layer_stream_k<false, 1>; break;
    default: break;
  }
  allow_big_lds((const void*)kern);
  LayerArgs a{(const char*)img, bias, ln, b_out, (const char*)ffn_img, b1, b2, g2, be2, x, stats_io, ldx, N, S, H, F / 32,
              1.4426950408889634f / sqrtf((float)dh)};
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, st, a);
}

void layer_phase_read(unsigned long long* out16) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_layer_phase), 16 * sizeof(unsigned long long));
  unsigned long long z[16] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_layer_phase), z, sizeof(z));
}

}  // namespace ldm
