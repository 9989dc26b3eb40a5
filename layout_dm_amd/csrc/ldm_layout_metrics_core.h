// Per-layout alignment / overlap metrics of the reference's evaluation (trainer/helpers/metric.py:98-203, called on the decoded
// boxes of every generated layout at eval.py:153-155,203-205) — ONE source of the per-element arithmetic, compiled for the
// device (kernels_metrics.hip: one wavefront per layout, lane = element) and for the host (tests/cpu_metrics_check.cpp).
//
// Reference semantics kept, quirks included:
//   compute_alignment (metric.py:98-149)
//     X = |v_k(i) - v_k(j)| over the six coordinates xl, xc, xr, yt, yc, yb (xl = xc - w / 2, ...: util.py:16-22), j over EVERY
//     slot of the padded layout — only the ROW of an invalid element is masked (X[~mask] = 1.0, l.113), so the stored boxes of
//     padded slots (zeros after decode / to_dense_batch) take part in a valid element's minimum; diagonal = 1; min over (k, j);
//     a minimum of exactly 1.0 becomes 0; term = -log(1 - min).   ACLayoutGAN = sum, LayoutGAN++ = sum / #valid (NaN -> 0).
//     NDN (l.126-141): min over xl, xc, xr and over VALID j != i of |v(j) - v(i)| for a valid i, 1.0 -> 0, summed.
//   compute_overlap (metric.py:152-203)
//     boxes of invalid elements are zeroed first; a1(i) = (r - l)(b - t); ai(i, j) = intersection area where l_max < r_min and
//     t_max < b_min, zero on the diagonal and where i or j is invalid; ar = nan_to_num(ai / a1(i)).
//     ACLayoutGAN = sum_ij ar, LayoutGAN++ = that / #valid (NaN -> 0), LayoutGAN = sum_{i<j} ai.
// fp32 throughout, like the reference (eval.py hands float32 boxes).  Sums run in index order (j inside i); torch's own
// reduction order is unspecified, so parity is to fp32 rounding (tests: rtol 1e-5), not bit-exact.
#pragma once
#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#define LDM_HD __host__ __device__ __forceinline__
#else
#define LDM_HD inline
#endif

namespace ldm_metrics {

constexpr int kNumMetrics = 6;  // out[0..5]: alignment-ACLayoutGAN, -LayoutGAN++, -NDN, overlap-ACLayoutGAN, -LayoutGAN++, -LayoutGAN

struct ElemTerms {
  float align;     // -log(1 - min |coordinate difference|) of this element (0 for an invalid one)
  float ndn;       // min |x-coordinate difference| to a valid other element (0 if none)
  float ar;        // sum_j nan_to_num(ai(i, j) / a1(i))
  float ai_upper;  // sum_{j > i} ai(i, j)
};

struct Box6 {
  float v[6];  // xl, xc, xr, yt, yc, yb
};
LDM_HD Box6 coords(const float* b) {
  const float hw = b[2] / 2, hh = b[3] / 2;
  return Box6{{b[0] - hw, b[0], b[0] + hw, b[1] - hh, b[1], b[1] + hh}};
}
LDM_HD float nan_to_num(float x) {
  if (x != x) return 0.f;
  if (x > FLT_MAX) return FLT_MAX;
  if (x < -FLT_MAX) return -FLT_MAX;
  return x;
}

// bbox: [S][4] (xc, yc, w, h) of ONE layout as stored (padded slots included), mask: [S] (1 = valid element)
template <typename MaskT>
LDM_HD ElemTerms element_terms(const float* bbox, const MaskT* mask, int S, int i) {
  ElemTerms t{0.f, 0.f, 0.f, 0.f};
  if (!mask[i]) return t;  // row masked: every term of an invalid element is 0 (-log(1 - 0) = 0)
  const Box6 ci = coords(bbox + 4 * i);
  float m = 1.0f, my = 1.0f;
  // overlap works on the zeroed copy of invalid boxes; i is valid here, so its own box is as stored
  const float l1 = ci.v[0], r1 = ci.v[2], t1 = ci.v[3], b1 = ci.v[5];
  const float a1 = (r1 - l1) * (b1 - t1);
  for (int j = 0; j < S; ++j) {
    if (j == i) continue;  // diagonal: 1.0 in both alignment forms, 0 in the overlap
    const Box6 cj = coords(bbox + 4 * j);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < 6; ++k) m = fminf(m, fabsf(ci.v[k] - cj.v[k]));
    if (mask[j]) {
      for (int k = 0; k < 3; ++k) my = fminf(my, fabsf(cj.v[k] - ci.v[k]));
      const float l_max = fmaxf(l1, cj.v[0]), r_min = fminf(r1, cj.v[2]);
      const float t_max = fmaxf(t1, cj.v[3]), b_min = fminf(b1, cj.v[5]);
      const float ai = (l_max < r_min && t_max < b_min) ? (r_min - l_max) * (b_min - t_max) : 0.f;
      t.ar += nan_to_num(ai / a1);
      if (j > i) t.ai_upper += ai;
    } else {
      // (an invalid j adds ai = 0: ar gains nan_to_num(0 / a1) — 0, or NaN -> 0 when a1 == 0)
    }
  }
  if (m == 1.0f) m = 0.f;
  t.align = -logf(1.0f - m);
  t.ndn = (my == 1.0f) ? 0.f : my;
  return t;
}

// the layout's six scores from its elements' terms, summed in index order
template <typename GetTerms>
LDM_HD void layout_scores(int S, int n_valid, GetTerms terms, float* out) {
  float sa = 0.f, sy = 0.f, so = 0.f, su = 0.f;
  for (int i = 0; i < S; ++i) {
    const ElemTerms t = terms(i);
    sa += t.align;
    sy += t.ndn;
    so += t.ar;
    su += t.ai_upper;
  }
  const float nv = (float)n_valid;
  float na = sa / nv, no = so / nv;
  if (na != na) na = 0.f;
  if (no != no) no = 0.f;
  out[0] = sa; out[1] = na; out[2] = sy; out[3] = so; out[4] = no; out[5] = su;
}

}  // namespace ldm_metrics
