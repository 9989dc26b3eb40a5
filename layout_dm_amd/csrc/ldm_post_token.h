// The tail of one reverse step for ONE token as straight-line scalar code on the token's sub-vocabulary — the form in
// which a single lane can run it behind the fused vocabulary head of the stack kernel (logits of a
// layout in LDS, one lane per token, no cross-lane traffic).  Same arithmetic, in the same order, as the wave-per-token
// kernel of kernels_post.hip (which stays the parity hook), restricted to the classes that can carry probability:
//
//   predict_start tail   log-softmax over the C-1 non-MASK classes, clamp [-70, 0]          base.py:131-144
//   q_posterior          body of the token's attribute + [PAD] + [MASK] (<= 34 classes for the reference's
//                        vocabularies; every other class sits at log(1e-30))               constrained.py:135-206
//   cond overrides       strong mask / refinement prior / [PAD] disable                     base.py:243-284
//   draw                 argmax | temperature, top-k, top-p, gumbel -> inverse CDF          helpers/sampling.py:81-130
//
// Dead classes (log 1e-30, i.e. 1e-30 of the probability mass each) are left out of the draw: against the full-vocabulary
// kernel that moves a CDF edge by < 1e-28 relative, below the fp64 resolution of the comparison.
//
// Pure C++ (no HIP types): compiled for the device by the kernels and for the host by tests/cpu_post_token_check.cpp,
// which runs it against the oracle on reference-produced states.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define LDM_PT_HD __host__ __device__ __forceinline__
#else
#define LDM_PT_HD inline
#endif

namespace ldm_post {

constexpr float kLogEps = -69.07755278982137f;  // log(1e-30): categorical_diffusion/util.py:8
constexpr int kMaxLive = 66;                    // body (<= 64 classes) + [PAD] + [MASK]

enum Sampler { kDeterministic = 0, kRandom = 1, kTopP = 2, kTopK = 3, kGumbel = 4 };  // LDM_SAMPLE_* of ldm_hip.h

// schedule scalars of one (attribute, t_post): rows of the [8][n_attr][T+1] device table (ldm_kernels.h ScheduleRow)
struct StepSchedule {
  float la, lb, lc;        // log_at, log_bt, log_ct                       at t
  float LA, LB, LC;        // log_cumprod_at / bt / ct                     at t
  float LAu, LBu, LCu;     // log_cumprod_at / bt / ct                     at u = (t - 1) mod (T + 1)
  float L1Cu;              // log_1_min_cumprod_ct                         at u
};

struct TokenArgs {
  const float* logits;     // [n_class] logits of this token (the [MASK] column is ignored)
  int tok;                 // x_t
  int start, count;        // body of the token's attribute: full ids start .. start + count - 1
  int pad_id, mask_id, n_class;
  // cond (base.py:243-284)
  int cond_tok;            // conditioned token or -1
  bool strong;             // strong mask set at this position
  const float* weak;       // refinement prior of this token, weak[c * weak_stride], or nullptr
  long weak_stride;
  bool pad_disable;        // cond type c / cwh / refinement / relation, attribute != 0, cond_tok != [PAD]
  // draw
  int kind;
  float temperature, top_p;
  int top_k;
  uint32_t pos, step;      // Philox counter words 0, 1
  uint64_t layout, seed;   // counter words 2, 3 (global layout index) and the key
};

LDM_PT_HD float log_add_exp(float a, float b) {  // util.py:19-21
  const float m = fmaxf(a, b);
  return m + logf(expf(a - m) + expf(b - m));
}

LDM_PT_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                             uint32_t (&out)[4]) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
LDM_PT_HD float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f; }  // strictly inside (0,1)

// full id of live slot i: the body in class order, then [PAD], then [MASK] (increasing ids: class order is kept)
LDM_PT_HD int live_id(const TokenArgs& a, int i) { return i < a.count ? a.start + i : (i == a.count ? a.pad_id : a.mask_id); }

// log p(x_{t-1} | x_t) of the live classes after the cond overrides -> lp[0 .. count + 1].  F64_LSE: the reference's
// float64 log-softmax (exact mode); otherwise fp32 (fast mode).
template <bool F64_LSE>
LDM_PT_HD void token_log_probs(const TokenArgs& a, const StepSchedule& s, float (&lp)[kMaxLive]) {
  const int K = a.count + 2, C = a.n_class;
  // ---- log-softmax over the C-1 non-MASK classes (all of them: the normaliser needs the dead ones too)
  float mx = -INFINITY;
  for (int c = 0; c < C - 1; ++c) mx = fmaxf(mx, a.logits[c]);
  float lse0f = 0.f;
  double lse0d = 0.0;
  if (F64_LSE) {
    double se = 0.0;
    for (int c = 0; c < C - 1; ++c) se += exp((double)a.logits[c] - (double)mx);
    lse0d = log(se);
  } else {
    float se = 0.f;
    for (int c = 0; c < C - 1; ++c) se += expf(a.logits[c] - mx);
    lse0f = logf(se);
  }
  auto log_x0 = [&](int c) {
    float v;
    if (c >= C - 1) v = -70.0f;
    else if (F64_LSE) v = (float)(((double)a.logits[c] - (double)mx) - lse0d);
    else v = (a.logits[c] - mx) - lse0f;
    return fminf(fmaxf(v, -70.0f), 0.0f);
  };
  // ---- constrained posterior on the live classes (constrained.py:166-197)
  const bool x_is_mask = a.tok == a.mask_id;
  float q[kMaxLive], q1[kMaxLive];
  float qmx = -INFINITY;
  for (int i = 0; i < K; ++i) {
    const int c = live_id(a, i);
    if (c == a.mask_id) {
      q[i] = kLogEps;                      // l.189
      q1[i] = x_is_mask ? 0.0f : kLogEps;  // l.179-185
    } else {
      float qt;
      if (x_is_mask) {
        qt = s.LC;  // l.169-173
        q1[i] = s.lc;
      } else {
        const float e = (c == a.tok) ? 0.0f : kLogEps;  // log one-hot of x_t (util.py:34-40)
        qt = log_add_exp(e + s.LA, s.LB);
        q1[i] = log_add_exp(e + s.la, s.lb);
      }
      q[i] = log_x0(c) - qt;  // l.188
    }
    qmx = fmaxf(qmx, q[i]);
  }
  float qs = 0.f;
  for (int i = 0; i < K; ++i) qs += expf(q[i] - qmx);
  const float lse = logf(qs) + qmx;  // torch.logsumexp
  for (int i = 0; i < K; ++i) {
    const int c = live_id(a, i);
    const float qn = q[i] - lse;
    const float r = (c == a.mask_id) ? log_add_exp(qn + s.L1Cu, s.LCu) : log_add_exp(qn + s.LAu, s.LBu);
    lp[i] = fminf(fmaxf((r + q1[i]) + lse, -70.0f), 0.0f);  // l.192-197
  }
  // ---- constraint injection (base.py:243-284)
  for (int i = 0; i < K; ++i) {
    const int c = live_id(a, i);
    if (a.strong) lp[i] = (c == a.cond_tok) ? 0.0f : kLogEps;
    else if (a.weak) lp[i] += a.weak[(long)c * a.weak_stride];
  }
  if (a.pad_disable) lp[a.count] = kLogEps;
}

// categorical draw over the live classes (helpers/sampling.py:81-130) -> full id
LDM_PT_HD int draw_live(const TokenArgs& a, const float (&lp)[kMaxLive]) {
  const int K = a.count + 2;
  if (a.kind == kDeterministic) {  // first maximum in class order
    int bi = 0;
    for (int i = 1; i < K; ++i)
      if (lp[i] > lp[bi]) bi = i;
    return live_id(a, bi);
  }
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const uint32_t l0 = (uint32_t)a.layout, l1 = (uint32_t)(a.layout >> 32);
  float lg[kMaxLive];
  for (int i = 0; i < K; ++i) lg[i] = lp[i] / a.temperature;
  if (a.kind == kGumbel) {  // noise per class: counter word 0 = pos | (1 + c / 4) << 16, component c & 3
    for (int i = 0; i < K; ++i) {
      const int c = live_id(a, i);
      uint32_t r[4];
      philox4x32_10(a.pos | ((uint32_t)(1 + (c >> 2)) << 16), a.step, l0, l1, k0, k1, r);
      lg[i] += -logf(-logf(u01(r[c & 3]) + 1e-30f) + 1e-30f);
    }
  }
  if (a.kind == kTopP || a.kind == kTopK) {
    float m1 = -INFINITY;
    for (int i = 0; i < K; ++i) m1 = fmaxf(m1, lg[i]);
    float ex[kMaxLive], es = 0.f;
    for (int i = 0; i < K; ++i) {
      ex[i] = expf(lg[i] - m1);
      es += ex[i];
    }
    // position in the descending (stable) order and the inclusive cumulative probability up to it
    int rank[kMaxLive];
    float cum[kMaxLive];
    for (int i = 0; i < K; ++i) {
      rank[i] = 0;
      cum[i] = 0.f;
      for (int o = 0; o < K; ++o) {
        const bool before = (lg[o] > lg[i]) || (lg[o] == lg[i] && o < i);
        if (before) {
          rank[i] += 1;
          cum[i] += ex[o] / es;
        } else if (o == i) {
          cum[i] += ex[o] / es;
        }
      }
    }
    if (a.kind == kTopP) {  // drop every class whose inclusive cumulative probability exceeds p, except the first
      for (int i = 0; i < K; ++i)
        if (cum[i] > a.top_p && rank[i] > 0) lg[i] = -INFINITY;
    } else {  // threshold = k-th largest value (sampling.py:73-78); k beyond the live classes keeps all of them
      float thr = INFINITY;
      for (int i = 0; i < K; ++i)
        if (rank[i] < a.top_k) thr = fminf(thr, lg[i]);
      for (int i = 0; i < K; ++i)
        if (lg[i] < thr) lg[i] = -INFINITY;
    }
  }
  // softmax -> inverse CDF in class order (the normaliser cancels: compare against u * total)
  float m2 = -INFINITY;
  for (int i = 0; i < K; ++i) m2 = fmaxf(m2, lg[i]);
  double cdf[kMaxLive], base = 0.0;
  for (int i = 0; i < K; ++i) {
    base += (double)expf(lg[i] - m2);
    cdf[i] = base;
  }
  uint32_t r[4];
  philox4x32_10(a.pos, a.step, l0, l1, k0, k1, r);
  const double thr = (double)u01(r[0]) * base;
  int n = 0;
  for (int i = 0; i < K; ++i)
    if (cdf[i] <= thr) n += 1;
  return live_id(a, n < K - 1 ? n : K - 1);
}

template <bool F64_LSE>
LDM_PT_HD int step_token(const TokenArgs& a, const StepSchedule& s) {
  float lp[kMaxLive];
  token_log_probs<F64_LSE>(a, s, lp);
  return draw_live(a, lp);
}

}  // namespace ldm_post
