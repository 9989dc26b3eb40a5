// Row kernels for gfx950: token+positional embedding gather, (Ada)LayerNorm, dtype casts and the
// load-time tables.  One 64-lane wavefront per row, float4 (16 B/lane) coalesced reads, wave-level
// DPP/shuffle reductions — no LDS, no cross-wave traffic.
//
// Reference semantics:
//   embedding     trainer/models/common/nn_lib.py:204,220 + ElementPositionalEmbedding 112-127
//   AdaLayerNorm  trainer/models/transformer_utils.py:72-83  (LN without affine, eps 1e-5)
//   LayerNorm     torch.nn.LayerNorm(d, eps=1e-5) (transformer_utils.py:156, nn_lib.py:186-189)
#include "ldm_kernels.h"

namespace ldm {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr float kLoScale = kSplitLoScale;  // split mode: lo = (x - fp16(x)) * kSplitLoScale (ldm_kernels.h)

// NV = float4 per lane (D <= 256*NV)
template <int NV>
__global__ __launch_bounds__(256) void ln_rows(LnArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const int nvec = a.D >> 2;
  float4 v[NV];
  if (a.tokens) {
    const int tok = a.tokens[row];
    const float4* e = reinterpret_cast<const float4*>(a.emb + (size_t)tok * a.D);
    const float4* p = reinterpret_cast<const float4*>(a.pos + (size_t)(row % a.S) * a.D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 64 + lane;
      if (c < nvec) {
        float4 x = e[c], y = p[c];
        v[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
      } else {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  } else {
    const float4* x = reinterpret_cast<const float4*>(a.x + (size_t)row * a.D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 64 + lane;
      v[i] = (c < nvec) ? x[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float inv_d = 1.0f / (float)a.D;
  const float mean = wave_sum(s) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 64 + lane;
    if (c < nvec) {
      float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float var = wave_sum(q) * inv_d;  // biased variance
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  if (a.stats_out && lane == 0) a.stats_out[row] = make_float2(mean, rstd);
  if (a.raw) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 64 + lane;
      if (c < nvec) reinterpret_cast<float4*>(a.y32 + (size_t)row * a.D)[c] = v[i];
    }
    return;
  }
  const float4* p0 = reinterpret_cast<const float4*>(a.p0);
  const float4* p1 = reinterpret_cast<const float4*>(a.p1);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 64 + lane;
    if (c >= nvec) continue;
    float4 g = p0[c], b = p1[c];
    float4 y;
    if (a.ada) {
      y.x = (v[i].x - mean) * rstd * (1.0f + g.x) + b.x;
      y.y = (v[i].y - mean) * rstd * (1.0f + g.y) + b.y;
      y.z = (v[i].z - mean) * rstd * (1.0f + g.z) + b.z;
      y.w = (v[i].w - mean) * rstd * (1.0f + g.w) + b.w;
    } else {
      y.x = (v[i].x - mean) * rstd * g.x + b.x;
      y.y = (v[i].y - mean) * rstd * g.y + b.y;
      y.z = (v[i].z - mean) * rstd * g.z + b.z;
      y.w = (v[i].w - mean) * rstd * g.w + b.w;
    }
    if (a.y32) reinterpret_cast<float4*>(a.y32 + (size_t)row * a.D)[c] = y;
    if (a.y16) {
      __half2 h0 = __floats2half2_rn(y.x, y.y), h1 = __floats2half2_rn(y.z, y.w);
      uint2 pk;
      pk.x = *reinterpret_cast<unsigned*>(&h0);
      pk.y = *reinterpret_cast<unsigned*>(&h1);
      reinterpret_cast<uint2*>(a.y16 + (size_t)row * a.ld16)[c] = pk;
      if (a.y16lo) {
        float2 f0 = __half22float2(h0), f1 = __half22float2(h1);
        __half2 l0 = __floats2half2_rn((y.x - f0.x) * kLoScale, (y.y - f0.y) * kLoScale);
        __half2 l1 = __floats2half2_rn((y.z - f1.x) * kLoScale, (y.w - f1.y) * kLoScale);
        pk.x = *reinterpret_cast<unsigned*>(&l0);
        pk.y = *reinterpret_cast<unsigned*>(&l1);
        reinterpret_cast<uint2*>(a.y16lo + (size_t)row * a.ld16)[c] = pk;
      }
    }
  }
}

void launch_layernorm(const LnArgs& a, hipStream_t st) {
  const int blocks = (a.M + 3) / 4;
  if (a.D <= 256) {
    hipLaunchKernelGGL(ln_rows<1>, dim3(blocks), dim3(256), 0, st, a);
  } else if (a.D <= 512) {
    hipLaunchKernelGGL(ln_rows<2>, dim3(blocks), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(ln_rows<4>, dim3(blocks), dim3(256), 0, st, a);
  }
}

// ------------------------------------------------------------------ casts
__global__ __launch_bounds__(256) void cast_f32_f16(const float* __restrict__ src, __half* __restrict__ dst,
                                                    __half* __restrict__ dstlo, int64_t n, float scale) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    const float x = src[i] * scale;  // (scale: an exact power of two — the split mode's weight pre-scale — or 1)
    const __half h = __float2half_rn(x);
    dst[i] = h;
    if (dstlo) dstlo[i] = __float2half_rn((x - __half2float(h)) * kLoScale);
  }
}

void launch_f32_to_f16(const float* src, __half* dst, __half* dstlo, int64_t n, hipStream_t st, float scale) {
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cast_f32_f16, dim3(blocks), dim3(256), 0, st, src, dst, dstlo, n, scale);
}

// ------------------------------------------------------------------ load-time tables
// out[t][layer][j] = b[j] + sum_d W[j][d] * silu(emb[t][d])   — one wave per (t, j)
__global__ __launch_bounds__(256) void adaln_table_k(const float* __restrict__ emb, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ out, int T,
                                                     int D, int L, int layer) {
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int two_d = 2 * D;
  if (idx >= T * two_d) return;
  const int t = idx / two_d, j = idx % two_d;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float e = emb[(size_t)t * D + d];
    const float sl = e / (1.0f + expf(-e));  // SiLU
    s += w[(size_t)j * D + d] * sl;
  }
  s = wave_sum(s);
  if (lane == 0) out[((size_t)t * L + layer) * two_d + j] = s + b[j];
}

void launch_adaln_table(const float* emb, const float* w, const float* b, float* out, int T, int D, int L, int layer,
                        hipStream_t st) {
  const int n = T * 2 * D;
  hipLaunchKernelGGL(adaln_table_k, dim3((n + 3) / 4), dim3(256), 0, st, emb, w, b, out, T, D, L, layer);
}

__global__ __launch_bounds__(256) void pos_table_k(const float* __restrict__ elem, const float* __restrict__ attr,
                                                   float* __restrict__ pos, int E, int A, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= E * A * D) return;
  const int s = i / D, d = i % D;
  pos[i] = elem[(size_t)(s / A) * D + d] + attr[(size_t)(s % A) * D + d];
}

void launch_pos_table(const float* elem, const float* attr, float* pos, int E, int A, int D, hipStream_t st) {
  const int n = E * A * D;
  hipLaunchKernelGGL(pos_table_k, dim3((n + 255) / 256), dim3(256), 0, st, elem, attr, pos, E, A, D);
}

// Lane phase offset: one wave spins for `us` microseconds on the constant 100 MHz s_memrealtime counter.
__global__ void delay_k(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
void launch_delay_us(int us, hipStream_t st) {
  hipLaunchKernelGGL(delay_k, dim3(1), dim3(64), 0, st, (unsigned long long)us * 100ull);
}

}  // namespace ldm
