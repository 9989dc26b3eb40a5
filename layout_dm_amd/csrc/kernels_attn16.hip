// MFMA self-attention for the fast numerics mode: softmax(Q K^T / sqrt(dh)) V per (layout, head),
// no mask (torch.nn.MultiheadAttention, trainer/models/transformer_utils.py:140-142,197-204).
// S = 125 (<=128) tokens and dh = 58 (<=64): one (layout, head) problem is ONE 128x128 score tile,
// so there is no online-softmax loop — a workgroup of 4 waves handles one (layout, head), wave w
// owning query rows 32w..32w+31.
//
// Layout contract (set up by the host, see ldm_weights.cpp): qkv is [M, 3*H*64] fp16 with every head's
// q/k/v slice padded 58 -> 64 columns (exact zeros: the padded in_proj rows/bias are zero), so each
// head row is one aligned 128-byte line; the output is [M, H*64] in the same head-padded layout and
// the out-projection weight has matching zero columns.
//
// gfx950 mapping: v_mfma_f32_32x32x16_f16 with swapped operands everywhere —
//   S^T tile = K_tile · Q_w^T  -> D[i = key][j = query]: lane (j = lane&31) holds 64 scores of ITS
//       query (the other 64 sit in lane^32), so the row softmax is in-register + one lane^32 exchange;
//   O^T = V^T · P^T            -> the P operand is exactly the registers the lane already holds
//       (k-slot order = accumulator order; V^T is read from LDS in the same order), and the result
//       D[i = d][j = query] gives each lane 4 consecutive d of its query: 8-byte stores.
// K is staged in LDS with the XOR-swizzled 128-B-row image (conflict-free ds_read_b128), V is staged
// transposed ([d][key], row stride 132 halfs -> conflict-free ds_read_b64).
#include <cstdlib>

#include "ldm_kernels.h"

namespace ldm {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

constexpr int VT_LD = 132;  // halfs per V^T row

// ABL (timing ablations only): 0 = full kernel, 1 = no global loads, 2 = loads + QK^T only, 3 = no PV
template <int ABL>
__global__ __launch_bounds__(256) void attn_mfma_k(const __half* __restrict__ qkv, __half* __restrict__ out, int S,
                                                   int H, int ldq, int ldo, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) __half Ks[128 * 64];
  __shared__ __attribute__((aligned(16))) __half Qs[128 * 64];
  __shared__ __attribute__((aligned(16))) __half Vt[64 * VT_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const size_t row0 = (size_t)b * S;
  const __half* qbase = qkv + row0 * ldq + (size_t)h * 64;
  const __half* kbase = qbase + (size_t)H * 64;
  const __half* vbase = qbase + (size_t)2 * H * 64;
  const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);

  {  // Q, K, V -> LDS with FULL-LINE loads: 8 consecutive lanes fetch the 8 16-B chunks of one 128-B head
     // row (one request per line instead of 4-8), 32 rows per pass.  Q/K keep the XOR-swizzled row image,
     // V is transposed on the way in (ds_write_b16).
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const bool ok = ABL != 1 && row < S;
      const size_t go = (size_t)row * ldq + c * 8;
      const uint4 kv = ok ? *reinterpret_cast<const uint4*>(kbase + go) : z4;
      const uint4 qv = ok ? *reinterpret_cast<const uint4*>(qbase + go) : z4;
      const uint4 vv = ok ? *reinterpret_cast<const uint4*>(vbase + go) : z4;
      const int so = row * 64 + ((c ^ ((row >> 1) & 7)) << 3);
      *reinterpret_cast<uint4*>(&Ks[so]) = kv;
      *reinterpret_cast<uint4*>(&Qs[so]) = qv;
      const unsigned short* pv = reinterpret_cast<const unsigned short*>(&vv);
      unsigned short* vt = reinterpret_cast<unsigned short*>(Vt);
#pragma unroll
      for (int e = 0; e < 8; ++e) vt[(c * 8 + e) * VT_LD + row] = pv[e];
    }
  }
  const int j = lane & 31;  // this lane's query (within the wave's 32) / fragment row
  const int hi = lane >> 5;
  const int q = wave * 32 + j;
  const int ksw = (j >> 1) & 7;
  __syncthreads();
  f16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = *reinterpret_cast<const f16x8*>(&Qs[q * 64 + (((ks * 2 + hi) ^ ksw) << 3)]);

  // ---- scores^T: 4 key tiles x 4 k-steps
  f32x16 sc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f16x8 kf = *reinterpret_cast<const f16x8*>(&Ks[(kt * 32 + j) * 64 + (((ks * 2 + hi) ^ ksw) << 3)]);
      sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sc[kt], 0, 0, 0);
    }
  }
  if (ABL == 2) {
    float t = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += sc[kt][r];
    if (q < S) out[(row0 + q) * ldo + (size_t)h * 64 + hi] = __float2half(t);
    return;
  }
  // ---- softmax over the 128 keys of query j (64 here, 64 in lane^32).  VALU-lean: with the reference's S = 125
  // only the last key tile holds padded keys (the wave-uniform test skips the other tiles; shorter sequences mask
  // them too); exp(scale*(s-max)) is one v_fma + one raw v_exp_f32 per score.
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt == 3 || S < (kt + 1) * 32) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
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
  const float nmxs = -mx * scale_log2e;
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], scale_log2e, nmxs));
      sc[kt][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;

  if (ABL == 3) {
    if (q < S) out[(row0 + q) * ldo + (size_t)h * 64 + hi] = __float2half(inv + (float)sc[1][3]);
    return;
  }
  // ---- O^T = V^T · P^T: 2 d-tiles x 8 k-steps (k-slot e of group hi <-> accumulator reg 8*half+e)
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      f16x8 pf;
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) pf[e2] = (_Float16)sc[kt][hf * 8 + e2];
      const int kb = kt * 32 + hf * 16 + hi * 4;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const __half* vr = &Vt[(dt * 32 + j) * VT_LD + kb];
        const f16x4 lo = *reinterpret_cast<const f16x4*>(vr);
        const f16x4 up = *reinterpret_cast<const f16x4*>(vr + 8);
        f16x8 vf;
        vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
        vf[4] = up[0]; vf[5] = up[1]; vf[6] = up[2]; vf[7] = up[3];
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[dt], 0, 0, 0);
      }
    }
  }
  if (q < S) {
    __half* orow = out + (row0 + q) * ldo + (size_t)h * 64;
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
}

void launch_attention16(const __half* qkv, __half* out, int B, int S, int H, int dh, int ldq, int ldo, hipStream_t st) {
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)dh);
  static const int abl = knob_int("LDM_ATTN_ABL", 0);  // (dev mode only: ldm_knobs.h)
  auto kern = abl == 1 ? attn_mfma_k<1> : abl == 2 ? attn_mfma_k<2> : abl == 3 ? attn_mfma_k<3> : attn_mfma_k<0>;
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(256), 0, st, qkv, out, S, H, ldq, ldo, scale_log2e);
}

}  // namespace ldm
