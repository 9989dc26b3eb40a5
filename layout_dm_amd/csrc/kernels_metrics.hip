// Alignment / overlap metrics of generated layouts on the device (SURVEY section 8f-1's remaining host metrics;
// trainer/helpers/metric.py:98-203 compute_alignment / compute_overlap, called per generated batch at eval.py:153-155,203-205).
// Fed by decode_layouts_k's output (boxes + validity mask resident in HBM): one wavefront per layout, lane = element,
// every lane scans the layout's S boxes from LDS (S <= 256; 25 for the reference's datasets), the six scores are summed in
// index order by lane 0.  HBM-bound and tiny: 17 B in per element, 24 B out per layout.  The arithmetic is the one source
// of ldm_layout_metrics_core.h (also compiled for the host: tests/cpu_metrics_check.cpp).
#include "ldm_kernels.h"
#include "ldm_layout_metrics_core.h"

namespace ldm {

constexpr int kMetricsMaxS = 256;

__global__ __launch_bounds__(64) void layout_metrics_k(const float* __restrict__ bbox, const uint8_t* __restrict__ mask, int S,
                                                       float* __restrict__ out) {
  __shared__ float sbox[kMetricsMaxS * 4];
  __shared__ uint8_t smask[kMetricsMaxS];
  __shared__ ldm_metrics::ElemTerms sterm[kMetricsMaxS];
  const int b = blockIdx.x, lane = threadIdx.x;
  const float4* gb = reinterpret_cast<const float4*>(bbox) + (size_t)b * S;
  int nv = 0;
  for (int i = lane; i < S; i += 64) {
    reinterpret_cast<float4*>(sbox)[i] = gb[i];
    const uint8_t m = mask[(size_t)b * S + i];
    smask[i] = m;
    nv += m ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nv += __shfl_xor(nv, o, 64);
  __syncthreads();
  for (int i = lane; i < S; i += 64) sterm[i] = ldm_metrics::element_terms(sbox, smask, S, i);
  __syncthreads();
  if (lane == 0)
    ldm_metrics::layout_scores(S, nv, [&](int i) { return sterm[i]; }, out + (size_t)b * ldm_metrics::kNumMetrics);
}

int launch_layout_metrics(const float* bbox, const uint8_t* mask, int B, int S, float* out, hipStream_t st) {
  if (S < 1 || S > kMetricsMaxS) return -1;
  if (B <= 0) return 0;
  hipLaunchKernelGGL(layout_metrics_k, dim3(B), dim3(64), 0, st, bbox, mask, S, out);
  return 0;
}

}  // namespace ldm
