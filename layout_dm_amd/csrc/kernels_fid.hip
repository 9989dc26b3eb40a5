// FIDNetV3.extract_features (trainer/fid/model.py:123-164) — the layout feature extractor behind the FID /
// precision-recall metrics of the reference's eval step, as ONE kernel: a workgroup per layout.
//
//   x_e   = relu(enc_fc_in([fc_bbox(bbox_e) | emb_label[label_e]]))            e = 0..N-1          (l.147-150)
//   X     = [token ; x_0 .. x_{N-1}]                                             (TransformerWithToken, l.26-37)
//   4 x nn.TransformerEncoderLayer(d=256, heads=4, ff=128, post-norm, ReLU):     (l.16-23; torch defaults)
//        X = LN1(X + MHA(X, key_padding_mask));  X = LN2(X + W2 relu(W1 X + b1) + b2)
//   feature = X[0]                                                               (l.152)
//
// Everything is fp32 FMA (the reference runs this net in fp32 and FID compares covariances of these features, so
// no reduced precision here); a layout is 26 x 256 activations: the whole network state lives in LDS, weights
// (transposed once on the host so that thread n reads W^T[k][n] coalesced) stream from L2.  The work is ~70 MFLOP
// per layout — three orders of magnitude below the sampling loop that produced the layouts.
#include "ldm_kernels.h"

namespace ldm {

constexpr int FID_D = 256;      // d_model
constexpr int FID_MAXS = 32;    // max tokens per layout (N + 1 <= 32)

// out[s][n] (n in [n0, n0+cnt)) = act(bias[n] + sum_k in[s][k] * Wt[k][n]);  one output column per thread
template <int K, bool RELU>
__device__ __forceinline__ void col_linear(const float* __restrict__ Wt, int ldw, const float* __restrict__ bias,
                                           const float* in, int ldi, float* out, int ldo, int S, int n, bool active) {
  float acc[FID_MAXS];
#pragma unroll
  for (int s = 0; s < FID_MAXS; ++s) acc[s] = 0.f;
  if (active) {
    for (int k = 0; k < K; ++k) {
      const float w = Wt[(size_t)k * ldw + n];
#pragma unroll
      for (int s = 0; s < FID_MAXS; ++s) acc[s] = fmaf(in[s * ldi + k], w, acc[s]);
    }
    const float b = bias[n];
#pragma unroll
    for (int s = 0; s < FID_MAXS; ++s) {
      if (s < S) {
        const float v = acc[s] + b;
        out[s * ldo + n] = RELU ? fmaxf(v, 0.f) : v;
      }
    }
  }
}

// rows of x (S x 256) <- LayerNorm(x + y) * g + b ; one wave per row (4 waves)
__device__ __forceinline__ void add_layernorm(float* x, const float* y, const float* g, const float* b, int S, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  for (int s = wave; s < S; s += 4) {
    float v[4], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = x[s * FID_D + lane + 64 * j] + y[s * FID_D + lane + 64 * j];
      sum += v[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / FID_D;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) var += (v[j] - mean) * (v[j] - mean);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    const float rstd = 1.0f / sqrtf(var / FID_D + 1e-5f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = lane + 64 * j;
      x[s * FID_D + n] = (v[j] - mean) * rstd * g[n] + b[n];
    }
  }
}

__global__ __launch_bounds__(256) void fid_features_k(FidArgs a) {
  extern __shared__ float sm[];
  float* x = sm;                          // [32][256] residual stream
  float* t0 = x + FID_MAXS * FID_D;       // [32][192] q|k|v of a head | [32][128] FFN hidden
  float* sc = t0 + FID_MAXS * 192;        // [32][33]  attention probabilities of a head
  float* y = sc + FID_MAXS * 33;          // [32][256] out-projection / FFN output (pre-residual)
  float* ao = y + FID_MAXS * FID_D;       // [32][256] concatenated head outputs
  float* cat = y;                         // [32][512] embedding concat (y | ao, before the first layer)
  // (no static __shared__ here: static + the 160 KiB dynamic maximum requested by allow_big_lds would exceed the CU's
  //  LDS and make hipFuncSetAttribute fail)
  int* s_keep = reinterpret_cast<int*>(ao + FID_MAXS * FID_D + 512);  // [32] 1 = token may be attended to
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int N = a.N, S = N + 1;
  // The column kernels below accumulate over all FID_MAXS rows whatever S is (their results for rows >= S are
  // discarded): zero every activation buffer once so that those rows are defined values, not whatever the LDS held.
  for (int i = tid; i < (int)((ao + FID_MAXS * FID_D + 512) - sm); i += 256) sm[i] = 0.f;
  __syncthreads();

  // ---- embedding: [fc_bbox(bbox) | emb_label(label)] -> enc_fc_in -> relu   (rows 1..N), row 0 = token
  for (int i = tid; i < N * 512; i += 256) {
    const int e = i / 512, c = i % 512;
    float v;
    if (c < FID_D) {
      const float* bb = a.bbox + ((size_t)b * N + e) * 4;
      v = a.fc_bbox_b[c];
#pragma unroll
      for (int k = 0; k < 4; ++k) v = fmaf(bb[k], a.fc_bbox_wt[k * FID_D + c], v);
    } else {
      long lab = a.label[(size_t)b * N + e];
      lab = lab < 0 ? 0 : (lab >= a.num_label ? a.num_label - 1 : lab);
      v = a.emb_label[(size_t)lab * FID_D + (c - FID_D)];
    }
    cat[(e + 1) * 512 + c] = v;
  }
  if (tid < FID_MAXS) s_keep[tid] = (tid == 0) ? 1 : (tid <= N ? (a.padding_mask[(size_t)b * N + tid - 1] ? 0 : 1) : 0);
  __syncthreads();
  col_linear<512, true>(a.fc_in_wt, FID_D, a.fc_in_b, cat + 512, 512, x + FID_D, FID_D, N, tid, true);
  x[tid] = a.token[tid];
  __syncthreads();

  const float scale = 0.125f;  // 1 / sqrt(64)
  for (int l = 0; l < a.n_layer; ++l) {
    const FidLayer& L = a.layer[l];
    // ---- multi-head self-attention, one head at a time
    for (int h = 0; h < 4; ++h) {
      // q | k | v of head h: 192 columns; in_proj rows (which*256 + h*64 + d)
      {
        const int which = tid / 64, d = tid % 64;
        const bool act = tid < 192;
        const int n = act ? which * FID_D + h * 64 + d : 0;
        float acc[FID_MAXS];
#pragma unroll
        for (int s = 0; s < FID_MAXS; ++s) acc[s] = 0.f;
        if (act) {
          for (int k = 0; k < FID_D; ++k) {
            const float w = L.in_wt[(size_t)k * 768 + n];
#pragma unroll
            for (int s = 0; s < FID_MAXS; ++s) acc[s] = fmaf(x[s * FID_D + k], w, acc[s]);
          }
          const float bq = L.in_b[n];
#pragma unroll
          for (int s = 0; s < FID_MAXS; ++s)
            if (s < S) t0[s * 192 + tid] = acc[s] + bq;
        }
      }
      __syncthreads();
      // scores + softmax: one (query, key) pair per thread pass; S <= 32 -> S*S <= 1024
      for (int i = tid; i < S * FID_MAXS; i += 256) {
        const int q = i / FID_MAXS, kk = i % FID_MAXS;
        float v = -INFINITY;
        if (kk < S && s_keep[kk]) {
          float d = 0.f;
          for (int c = 0; c < 64; ++c) d = fmaf(t0[q * 192 + c], t0[kk * 192 + 64 + c], d);
          v = d * scale;
        }
        sc[q * 33 + kk] = v;
      }
      __syncthreads();
      if (tid < S) {
        float mx = -INFINITY;
        for (int kk = 0; kk < S; ++kk) mx = fmaxf(mx, sc[tid * 33 + kk]);
        float sum = 0.f;
        for (int kk = 0; kk < S; ++kk) {
          const float p = expf(sc[tid * 33 + kk] - mx);
          sc[tid * 33 + kk] = p;
          sum += p;
        }
        const float inv = 1.0f / sum;
        for (int kk = 0; kk < S; ++kk) sc[tid * 33 + kk] *= inv;
      }
      __syncthreads();
      // head output: ao[q][h*64 + d] = sum_k p[q][k] v[k][d]
      for (int i = tid; i < S * 64; i += 256) {
        const int q = i / 64, d = i % 64;
        float o = 0.f;
        for (int kk = 0; kk < S; ++kk) o = fmaf(sc[q * 33 + kk], t0[kk * 192 + 128 + d], o);
        ao[q * FID_D + h * 64 + d] = o;
      }
      __syncthreads();
    }
    col_linear<FID_D, false>(L.out_wt, FID_D, L.out_b, ao, FID_D, y, FID_D, S, tid, true);
    __syncthreads();
    add_layernorm(x, y, L.n1_g, L.n1_b, S, tid);
    __syncthreads();
    // ---- feed forward 256 -> 128 -> 256
    col_linear<FID_D, true>(L.w1t, 128, L.b1, x, FID_D, t0, 128, S, tid, tid < 128);
    __syncthreads();
    col_linear<128, false>(L.w2t, FID_D, L.b2, t0, 128, y, FID_D, S, tid, true);
    __syncthreads();
    add_layernorm(x, y, L.n2_g, L.n2_b, S, tid);
    __syncthreads();
  }
  a.feat[(size_t)b * FID_D + tid] = x[tid];
}

void launch_fid_features(const FidArgs& a, int B, hipStream_t st) {
  // (+512: the embedding GEMM reads one row past the 32-row concat buffer for the padded token slots)
  const size_t lds = (size_t)(3 * FID_MAXS * FID_D + FID_MAXS * 192 + FID_MAXS * 33 + 512 + 64) * sizeof(float);
  allow_big_lds((const void*)fid_features_k);
  hipLaunchKernelGGL(fid_features_k, dim3(B), dim3(256), lds, st, a);
}

}  // namespace ldm
