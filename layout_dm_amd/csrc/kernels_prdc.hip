// Precision / recall / density / coverage of two feature sets on the device — the k-NN half of
// compute_generative_model_scores (trainer/helpers/metric.py:37-59), which the reference delegates to the third-party
// package prdc (pyproject: prdc ^0.2; Naeem et al., "Reliable Fidelity and Diversity Metrics for Generative Models",
// ICML 2020).  prdc is not vendored in the reference tree; its published algorithm, restated:
//   radius_X[i]  = distance from X[i] to its nearest_k-th nearest neighbour inside X (the (nearest_k + 1)-th smallest entry
//                  of the row of the pairwise-distance matrix, whose smallest entry is the zero self-distance)
//   precision    = mean_j  any_i  d(real_i, fake_j) < radius_real[i]
//   recall       = mean_i  any_j  d(real_i, fake_j) < radius_fake[j]
//   density      = mean_j  sum_i [d(real_i, fake_j) < radius_real[i]] / nearest_k
//   coverage     = mean_i  min_j d(real_i, fake_j) < radius_real[i]
// Everything is compared on SQUARED distances (monotone).  These are HBM-bound reductions over n x m matrices of a few
// thousand rows — a metric that runs once per evaluation, not a hot path: straightforward kernels.
#include "ldm_kernels.h"

namespace ldm {

// D[i][j] = |A[i] - B[j]|^2, difference form (no cancellation), 32 x 32 output tile per workgroup
__global__ __launch_bounds__(256) void prdc_pdist2_k(const float* __restrict__ A, int n, const float* __restrict__ B, int m,
                                                     int dim, float* __restrict__ D) {
  __shared__ float As[32][33], Bs[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int d0 = 0; d0 < dim; d0 += 32) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q, d = d0 + tx;
      As[r][tx] = (i0 + r < n && d < dim) ? A[(size_t)(i0 + r) * dim + d] : 0.f;
      Bs[r][tx] = (j0 + r < m && d < dim) ? B[(size_t)(j0 + r) * dim + d] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int dd = 0; dd < 32; ++dd) {
      const float b = Bs[tx][dd];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float diff = As[ty + 8 * q][dd] - b;
        acc[q] = fmaf(diff, diff, acc[q]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (i0 + ty + 8 * q < n && j0 + tx < m) D[(size_t)(i0 + ty + 8 * q) * m + j0 + tx] = acc[q];
}

// r2[i] = the k1-th smallest entry of row i of D (k1 = nearest_k + 1 <= 8), one wavefront per row: every lane keeps
// the 8 smallest of its strided share in a sorted register list, then k1 rounds extract the wave-wide minimum
__global__ __launch_bounds__(256) void prdc_kth_k(const float* __restrict__ D, int n, int m, int k1, float* __restrict__ r2) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  float best[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) best[s] = INFINITY;
  for (int j = lane; j < m; j += 64) {
    float v = D[(size_t)row * m + j];
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (v < best[s]) {
        const float t = best[s];
        best[s] = v;
        v = t;
      }
  }
  float kth = INFINITY;
  for (int r = 0; r < k1; ++r) {
    float mn = best[0];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = fminf(mn, __shfl_xor(mn, off, 64));
    kth = mn;
    const unsigned long long owners = __ballot(best[0] == mn);
    if (lane == (int)__ffsll((long long)owners) - 1) {  // the first lane holding the minimum pops it
#pragma unroll
      for (int s = 0; s < 7; ++s) best[s] = best[s + 1];
      best[7] = INFINITY;
    }
  }
  if (lane == 0) r2[row] = kth;
}

// per real row i: recall_i (some fake inside that fake's radius) and coverage_i (nearest fake inside real_i's radius)
__global__ __launch_bounds__(256) void prdc_rows_k(const float* __restrict__ Drf, int n, int m, const float* __restrict__ r2_real,
                                                   const float* __restrict__ r2_fake, unsigned long long* __restrict__ counts) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  bool rec = false;
  float mn = INFINITY;
  for (int j = lane; j < m; j += 64) {
    const float d = Drf[(size_t)row * m + j];
    rec = rec || d < r2_fake[j];
    mn = fminf(mn, d);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mn = fminf(mn, __shfl_xor(mn, off, 64));
  const bool any_rec = __ballot(rec) != 0ull;
  if (lane == 0) {
    if (any_rec) atomicAdd(&counts[1], 1ull);
    if (mn < r2_real[row]) atomicAdd(&counts[3], 1ull);
  }
}
// per fake column j: precision_j (inside some real's radius) and its density count
__global__ __launch_bounds__(256) void prdc_cols_k(const float* __restrict__ Drf, int n, int m, const float* __restrict__ r2_real,
                                                   unsigned long long* __restrict__ counts) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  unsigned inside = 0;
  for (int i = 0; i < n; ++i) inside += Drf[(size_t)i * m + j] < r2_real[i] ? 1u : 0u;
  if (inside) atomicAdd(&counts[0], 1ull);
  atomicAdd(&counts[2], (unsigned long long)inside);  // (density sums up to n_real x n_fake: 64-bit)
}

void launch_prdc_pdist2(const float* A, int n, const float* B, int m, int dim, float* D, hipStream_t st) {
  hipLaunchKernelGGL(prdc_pdist2_k, dim3((m + 31) / 32, (n + 31) / 32), dim3(256), 0, st, A, n, B, m, dim, D);
}
void launch_prdc_kth(const float* D, int n, int m, int k1, float* r2, hipStream_t st) {
  hipLaunchKernelGGL(prdc_kth_k, dim3((n + 3) / 4), dim3(256), 0, st, D, n, m, k1, r2);
}
void launch_prdc_counts(const float* Drf, int n, int m, const float* r2_real, const float* r2_fake, unsigned long long* counts,
                        hipStream_t st) {
  hipLaunchKernelGGL(prdc_rows_k, dim3((n + 3) / 4), dim3(256), 0, st, Drf, n, m, r2_real, r2_fake, counts);
  hipLaunchKernelGGL(prdc_cols_k, dim3((m + 255) / 256), dim3(256), 0, st, Drf, n, m, r2_real, counts);
}

}  // namespace ldm
