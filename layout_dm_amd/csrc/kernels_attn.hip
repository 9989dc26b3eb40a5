// Self-attention of the denoiser: softmax(Q K^T / sqrt(dh)) V per (layout, head), no mask
// (torch.nn.MultiheadAttention with key_padding_mask=None, attn_mask=None:
//  trainer/models/transformer_utils.py:140-142,197-204; nn_lib.py:226-228).
// S = 125 tokens, dh = 58: one (layout, head) problem is a single tile.
//
// This file holds the exact-fp32 kernel: one workgroup per (layout, head), K and V of that head
// staged once in LDS (2 x 125 x 60 x 4 B = 60 KB), one query row per lane, two passes
// (row max, then exp/accumulate) so the arithmetic has the same form as torch's softmax.
// All lanes read the same K/V row => LDS broadcast reads (ds_read_b128, no bank conflicts).
#include <cstdlib>
#include <string>

#include "ldm_kernels.h"

namespace ldm {

template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <>
__device__ __forceinline__ float ldf<float>(const float* p) {
  return *p;
}
template <>
__device__ __forceinline__ float ldf<__half>(const __half* p) {
  return __half2float(*p);
}

constexpr float kLoScaleA = kSplitLoScale;

template <typename TIn, int DHP>
__global__ __launch_bounds__(128) void attn_rows(const TIn* __restrict__ qkv, float* __restrict__ out32,
                                                 __half* __restrict__ out16, __half* __restrict__ out16lo, int S,
                                                 int H, int dh, int D, int ld, int ldo32, int ldo16, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                    // [S][DHP]
  float* Vs = smem + (size_t)S * DHP;  // [S][DHP]
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const size_t row0 = (size_t)b * S;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < S * DHP; idx += 128) {
    const int j = idx / DHP, d = idx % DHP;
    float kv = 0.f, vv = 0.f;
    if (d < dh) {
      const TIn* r = qkv + (row0 + j) * ld + h * dh + d;
      kv = ldf<TIn>(r + D);
      vv = ldf<TIn>(r + 2 * D);
    }
    Ks[idx] = kv;
    Vs[idx] = vv;
  }
  __syncthreads();
  for (int i = tid; i < S; i += 128) {
    float q[DHP];
    const TIn* qr = qkv + (row0 + i) * ld + h * dh;
#pragma unroll
    for (int d = 0; d < DHP; ++d) q[d] = (d < dh) ? ldf<TIn>(qr + d) * scale : 0.f;
    float mx = -INFINITY;
    for (int j = 0; j < S; ++j) {
      const float4* kr = reinterpret_cast<const float4*>(Ks + (size_t)j * DHP);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < DHP / 4; ++d4) {
        const float4 k4 = kr[d4];
        s0 = fmaf(q[4 * d4 + 0], k4.x, s0);
        s1 = fmaf(q[4 * d4 + 1], k4.y, s1);
        s2 = fmaf(q[4 * d4 + 2], k4.z, s2);
        s3 = fmaf(q[4 * d4 + 3], k4.w, s3);
      }
      mx = fmaxf(mx, (s0 + s1) + (s2 + s3));
    }
    float o[DHP];
#pragma unroll
    for (int d = 0; d < DHP; ++d) o[d] = 0.f;
    float l = 0.f;
    for (int j = 0; j < S; ++j) {
      const float4* kr = reinterpret_cast<const float4*>(Ks + (size_t)j * DHP);
      const float4* vr = reinterpret_cast<const float4*>(Vs + (size_t)j * DHP);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < DHP / 4; ++d4) {
        const float4 k4 = kr[d4];
        s0 = fmaf(q[4 * d4 + 0], k4.x, s0);
        s1 = fmaf(q[4 * d4 + 1], k4.y, s1);
        s2 = fmaf(q[4 * d4 + 2], k4.z, s2);
        s3 = fmaf(q[4 * d4 + 3], k4.w, s3);
      }
      const float p = expf(((s0 + s1) + (s2 + s3)) - mx);
      l += p;
#pragma unroll
      for (int d4 = 0; d4 < DHP / 4; ++d4) {
        const float4 v4 = vr[d4];
        o[4 * d4 + 0] = fmaf(p, v4.x, o[4 * d4 + 0]);
        o[4 * d4 + 1] = fmaf(p, v4.y, o[4 * d4 + 1]);
        o[4 * d4 + 2] = fmaf(p, v4.z, o[4 * d4 + 2]);
        o[4 * d4 + 3] = fmaf(p, v4.w, o[4 * d4 + 3]);
      }
    }
    const float inv = 1.0f / l;
    const size_t orow32 = (row0 + i) * ldo32 + h * dh;
    const size_t orow16 = (row0 + i) * ldo16 + h * dh;
#pragma unroll
    for (int d = 0; d < DHP; ++d) {
      if (d < dh) {
        const float v = o[d] * inv;
        if (out32) out32[orow32 + d] = v;
        if (out16) {
          const __half hv = __float2half_rn(v);
          out16[orow16 + d] = hv;
          if (out16lo) out16lo[orow16 + d] = __float2half_rn((v - __half2float(hv)) * kLoScaleA);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// fp32 MFMA attention (exact / split modes): v_mfma_f32_32x32x2_f32 — fp32 operands, fp32 accumulate, i.e. the same
// arithmetic class as the reference's fp32 bmm, at 64 FLOP/clk/SIMD instead of the VALU kernel above (which spent
// 29 % of the exact-mode step at 6 % of the fp32 peak).  One workgroup per (layout, head), wave w owns queries
// 32w..32w+31; S <= 128, dh <= 64: a single score tile per query block, no online softmax.
//   S^T[key][query] = K Q^T : A = K fragment from LDS, B = Q fragment held in registers (29 k2-steps)
//     -> lane (query, hi) holds keys (r&3) + 8(r>>2) + 4hi of every 32-key tile: softmax = in-lane + one lane^32 exchange
//   O^T[d][query]   = V^T P^T: k2-step r of key tile kt pairs key (r&3)+8(r>>2) (lanes hi=0) with key +4 (lanes hi=1):
//     accumulator register r of the score tile IS the B operand, A = V rows of those two keys from LDS.
using f32x16a = __attribute__((ext_vector_type(16))) float;

template <int DH2>  // DH2 = ceil(dh / 2) k2-steps of the score GEMM
__global__ __launch_bounds__(256) void attn32_mfma_k(const float* __restrict__ qkv, float* __restrict__ out32,
                                                    __half* __restrict__ out16, __half* __restrict__ out16lo, int S,
                                                    int H, int dh, int D, int ld, int ldo32, int ldo16, float scale) {
  constexpr int KP = 2 * DH2 + 1;   // K row stride (odd: conflict-free ds_read_b32 down a column)
  constexpr int VP = 65;            // V row stride (d tiles 0..63)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                 // [128][KP]
  float* Vs = smem + 128 * KP;      // [128][VP] (+ tail)
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const size_t row0 = (size_t)b * S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  // K and V of this (layout, head) -> LDS.  Every load of a tile is issued before the first ds_write (a loop of
  // dependent load / store pairs pays one L2 round trip per iteration: 63 of them were ~40 % of this kernel)
  {
    constexpr int NK = (128 * KP + 255) / 256, NV = (128 * VP + 64 + 255) / 256;
    float rk[NK], rv[NV];
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int idx = tid + 256 * i, key = idx / KP, d = idx % KP;
      rk[i] = (idx < 128 * KP && key < S && d < dh) ? qkv[(row0 + key) * ld + D + h * dh + d] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + 256 * i, key = idx / VP, d = idx % VP;
      rv[i] = (idx < 128 * VP + 64 && key < S && d < dh) ? qkv[(row0 + key) * ld + 2 * D + h * dh + d] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NK; ++i)
      if (tid + 256 * i < 128 * KP) Ks[tid + 256 * i] = rk[i];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (tid + 256 * i < 128 * VP + 64) Vs[tid + 256 * i] = rv[i];
  }
  // Q fragment: lane (query j, hi) holds Q[query][2ks + hi]
  const int q = wave * 32 + j;
  float qf[DH2];
  {
    const float* qr = qkv + (row0 + (q < S ? q : S - 1)) * ld + h * dh;
#pragma unroll
    for (int ks = 0; ks < DH2; ++ks) qf[ks] = (2 * ks + hi < dh) ? qr[2 * ks + hi] * scale : 0.f;
  }
  __syncthreads();
  f32x16a sc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) sc[kt][i] = 0.f;
    const float* kr = Ks + (kt * 32 + j) * KP + hi;
#pragma unroll
    for (int ks = 0; ks < DH2; ++ks) sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * ks], qf[ks], sc[kt], 0, 0, 0);
  }
  // softmax over the 128 keys of this lane's query (64 here, 64 in lane^32); keys >= S masked
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (key >= S) sc[kt][r] = -INFINITY;
      mx = fmaxf(mx, sc[kt][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = expf(sc[kt][r] - mx);
      sc[kt][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  f32x16a o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;  // this lane half's key of k2-step r
      const float* vr = Vs + key * VP + j;
      o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], sc[kt][r], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], sc[kt][r], o[1], 0, 0, 0);
    }
  if (q < S) {
    // lane (query, hi) holds d = 32dt + (r&3) + 8(r>>2) + 4hi
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (d < dh) {
          const float v = o[dt][r] * inv;
          if (out32) out32[(row0 + q) * ldo32 + h * dh + d] = v;
          if (out16) {
            const __half hv = __float2half_rn(v);
            out16[(row0 + q) * ldo16 + h * dh + d] = hv;
            if (out16lo) out16lo[(row0 + q) * ldo16 + h * dh + d] = __float2half_rn((v - __half2float(hv)) * kLoScaleA);
          }
        }
      }
  }
}

// Second form of the same product (even dh, even leading dimensions: the reference's backbone): only K goes through
// LDS.  V is the A operand of O^T = V^T P^T with lane (d, key pair) <- V[key][d]: for a fixed k2-step the 32 lanes of a
// half read 32 consecutive floats of one row of qkv, a coalesced global load — so V is read straight into the operand
// registers, one 32-key tile ahead of the MFMAs that consume it (each wave reads the head's V once: 30 KB, L2 / TCP
// hits for three of the four).  The finished O^T is transposed through the K region and stored as whole rows (232
// contiguous bytes per query instead of 58 scattered dwords).  LDS 30 KB and <= 170 VGPRs: three workgroups per CU,
// whose load / MFMA / store phases overlap (the LDS-staged form above holds 64 KB: two).
template <int DH2>
__global__ __launch_bounds__(256, 3) void attn32_direct_k(const float* __restrict__ qkv, float* __restrict__ out32,
                                                         __half* __restrict__ out16, __half* __restrict__ out16lo, int S,
                                                         int H, int D, int ld, int ldo32, int ldo16, float scale) {
  constexpr int dh = 2 * DH2;
  constexpr int KP = dh + 1;  // K row stride (odd: conflict-free ds_read_b32 down a column)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;           // [128][KP]: K, then O (query-major)
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const size_t row0 = (size_t)b * S;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (an SGPR: nothing lane-derived has to live to the end)
  const int j = lane & 31, hi = lane >> 5;
  const int q = wave * 32 + j;
  // Q fragment: lane (query j, hi) holds Q[query][2ks + hi] (scattered dword loads, issued first: they land under the
  // K staging)
  float qf[DH2];
  {
    const float* qr = qkv + (row0 + (q < S ? q : S - 1)) * ld + h * dh + hi;
#pragma unroll
    for (int ks = 0; ks < DH2; ++ks) qf[ks] = qr[2 * ks];
  }
  // K of this (layout, head) -> LDS: 128 x DH2 float2, every load issued before the first ds_write
  {
    constexpr int NK = (128 * DH2 + 255) / 256;
    float2 rk[NK];
    const float* kb = qkv + row0 * ld + D + h * dh;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int idx = tid + 256 * i, key = idx / DH2, c = idx % DH2;
      rk[i] = (idx < 128 * DH2 && key < S) ? *reinterpret_cast<const float2*>(kb + (size_t)key * ld + 2 * c)
                                           : make_float2(0.f, 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int idx = tid + 256 * i, key = idx / DH2, c = idx % DH2;
      if (idx < 128 * DH2) {
        Ks[key * KP + 2 * c] = rk[i].x;
        Ks[key * KP + 2 * c + 1] = rk[i].y;
      }
    }
  }
  // V operand registers of key tile kt: k2-step r pairs key (r&3)+8(r>>2) (lanes hi=0) with key +4 (lanes hi=1), d tile dt
  // = columns 32dt + j.  Keys >= S and columns >= dh are clamped to valid addresses: their P is 0 / their O rows are
  // dropped (an MFMA output row depends on its own A row only).
  const float* vb = qkv + row0 * ld + 2 * D + h * dh;
  const int d0 = j, d1 = (32 + j < dh) ? 32 + j : dh - 1;
  // operands of half a key tile (8 k2-steps x 2 d tiles) per buffer, a ring of three: two half tiles (32 MFMAs of this
  // wave, ~100 of the SIMD's three) in flight ahead of the one being consumed
  float va[3][16];
  // addressing: wave-uniform row pointer (compile-time key of the k2-step) + one 32-bit lane offset per d tile, so a load
  // costs no address registers; only the last tile, which can run past S, clamps per lane
  const unsigned bo0 = 4u * (unsigned)(4 * hi * ld + d0), bo1 = 4u * (unsigned)(4 * hi * ld + d1);  // BYTE offsets (< 2^32)
  auto at = [](const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  auto vload = [&](int ht, float(&dst)[16]) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = (ht & 1) * 8 + rr;
      const int kc = (ht >> 1) * 32 + (r & 3) + 8 * (r >> 2);
      if ((ht >> 1) < 3) {  // (S > 96)
        const float* vr = vb + (size_t)kc * ld;
        dst[2 * rr] = at(vr, bo0);
        dst[2 * rr + 1] = at(vr, bo1);
      } else {
        int key = kc + 4 * hi;
        key = key < S ? key : S - 1;
        const unsigned ko = 4u * (unsigned)(key * ld);
        dst[2 * rr] = at(vb, ko + 4u * (unsigned)d0);
        dst[2 * rr + 1] = at(vb, ko + 4u * (unsigned)d1);
      }
    }
  };
  vload(0, va[0]);
  vload(1, va[1]);
#pragma unroll
  for (int ks = 0; ks < DH2; ++ks) qf[ks] *= scale;
  __syncthreads();
  f32x16a sc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) sc[kt][i] = 0.f;
    const float* kr = Ks + (kt * 32 + j) * KP + hi;
#pragma unroll
    for (int ks = 0; ks < DH2; ++ks) sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * ks], qf[ks], sc[kt], 0, 0, 0);
  }
  __syncthreads();  // every wave is done with K: the region becomes the O staging buffer
  // softmax over the 128 keys of this lane's query (64 here, 64 in lane^32); keys >= S masked
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (key >= S) sc[kt][r] = -INFINITY;
      mx = fmaxf(mx, sc[kt][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = expf(sc[kt][r] - mx);
      sc[kt][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  f32x16a o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
#pragma unroll
  for (int ht = 0; ht < 8; ++ht) {
    if (ht + 2 < 8) vload(ht + 2, va[(ht + 2) % 3]);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = (ht & 1) * 8 + rr;
      o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[ht % 3][2 * rr], sc[ht >> 1][r], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[ht % 3][2 * rr + 1], sc[ht >> 1][r], o[1], 0, 0, 0);
    }
  }
  // O^T -> LDS, query-major (lane (query, hi) holds d = 32dt + (r&3) + 8(r>>2) + 4hi), then whole rows to memory.
  // The lane id is re-read from the hardware here rather than kept alive across the MFMA phases.
  int lane2;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane2));
  const int j2 = lane2 & 31, hi2 = lane2 >> 5, q2 = wave * 32 + j2;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi2;
      if (d < dh) Ks[q2 * KP + d] = o[dt][r] * inv;
    }
  __syncthreads();
  constexpr int NO = (32 * DH2 + 63) / 64;
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int idx = lane2 + 64 * i, qr = wave * 32 + idx / DH2, c = idx % DH2;
    if (idx < 32 * DH2 && qr < S) {
      const float v0 = Ks[qr * KP + 2 * c], v1 = Ks[qr * KP + 2 * c + 1];
      if (out32) *reinterpret_cast<float2*>(out32 + (row0 + qr) * ldo32 + h * dh + 2 * c) = make_float2(v0, v1);
      if (out16) {
        const __half h0 = __float2half_rn(v0), h1 = __float2half_rn(v1);
        const size_t oo = (row0 + qr) * ldo16 + h * dh + 2 * c;
        *reinterpret_cast<__half2*>(out16 + oo) = __halves2half2(h0, h1);
        if (out16lo)
          *reinterpret_cast<__half2*>(out16lo + oo) = __halves2half2(__float2half_rn((v0 - __half2float(h0)) * kLoScaleA),
                                                                    __float2half_rn((v1 - __half2float(h1)) * kLoScaleA));
      }
    }
  }
}

template <typename TIn>
static void launch_rows(const AttnArgs& a, hipStream_t st) {
  const float scale = 1.0f / sqrtf((float)a.dh);
  const int dhp = (a.dh + 3) & ~3;
  float* o32 = a.out32;
  __half* o16 = a.out16;
  __half* o16lo = a.out16lo;
  dim3 grid(a.B * a.H), block(128);
#define LDM_ATTN_CASE(DHP)                                                                                      \
  {                                                                                                             \
    const size_t sh = (size_t)2 * a.S * DHP * sizeof(float);                                                    \
    allow_big_lds((const void*)attn_rows<TIn, DHP>);                                                            \
    hipLaunchKernelGGL((attn_rows<TIn, DHP>), grid, block, sh, st, (const TIn*)a.qkv, o32, o16, o16lo, a.S, a.H, \
                       a.dh, a.D, a.ld, a.ldo32, a.ldo16, scale);                                                          \
  }
  if (dhp <= 32) LDM_ATTN_CASE(32)
  else if (dhp <= 60) LDM_ATTN_CASE(60)
  else LDM_ATTN_CASE(64)
#undef LDM_ATTN_CASE
}

void launch_attention(const AttnArgs& a, hipStream_t st) {
  // LDM_ATTN32=rows selects the r01 VALU kernel for A/B timing
  static const std::string attn32_knob = knob_env("LDM_ATTN32") ? knob_env("LDM_ATTN32") : "";
  static const bool use_rows = attn32_knob == "rows";
  static const bool use_staged = attn32_knob == "staged";  // A/B timing
  // split mode: the fp16 x 3 kernel (LDM_ATTN32=direct keeps the fp32-MFMA kernel for A/B timing)
  if (!a.in_f16 && a.out16 && a.out16lo && !a.out32 && attn32_knob.empty() &&
      attention16x3_supported(a.S, a.dh, a.D, a.ld, a.ldo16)) {
    launch_attention16x3((const float*)a.qkv, a.out16, a.out16lo, a.B, a.S, a.H, a.dh, a.D, a.ld, a.ldo16, st);
    return;
  }
  if (!a.in_f16 && !use_rows && !use_staged && a.S <= 128 && a.S > 96 && a.dh == 58 && a.D % 2 == 0 && a.ld % 2 == 0 &&
      a.ldo32 % 2 == 0 && a.ldo16 % 2 == 0) {
    constexpr int DH2 = 29;
    const size_t sh = (size_t)(128 * (2 * DH2 + 1)) * sizeof(float);
    hipLaunchKernelGGL((attn32_direct_k<DH2>), dim3(a.B * a.H), dim3(256), sh, st, (const float*)a.qkv, a.out32, a.out16,
                       a.out16lo, a.S, a.H, a.D, a.ld, a.ldo32, a.ldo16, 1.0f / sqrtf((float)a.dh));
    return;
  }
  if (!a.in_f16 && !use_rows && a.S <= 128 && a.S > 96 && a.dh <= 58 && a.dh > 56) {
    constexpr int DH2 = 29;
    const size_t sh = (size_t)(128 * (2 * DH2 + 1) + 128 * 65 + 64) * sizeof(float);
    allow_big_lds((const void*)attn32_mfma_k<DH2>);
    hipLaunchKernelGGL((attn32_mfma_k<DH2>), dim3(a.B * a.H), dim3(256), sh, st, (const float*)a.qkv, a.out32, a.out16,
                       a.out16lo, a.S, a.H, a.dh, a.D, a.ld, a.ldo32, a.ldo16, 1.0f / sqrtf((float)a.dh));
    return;
  }
  if (a.in_f16)
    launch_rows<__half>(a, st);
  else
    launch_rows<float>(a, st);
}

}  // namespace ldm
