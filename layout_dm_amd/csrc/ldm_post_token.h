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

// Working storage: 3 * (count + 2) floats supplied by the caller (a lane of the kernel has no private arrays: indexed
// private memory would be scratch).  `work` MAY BE the token's own logits row (work == logits, row length >= 3 K): the
// log-softmax passes only read; the posterior then reads logits[live_id(i)] before it writes work[i], live_id(i) >= i,
// and everything behind that lives in work alone.
//
// log p(x_{t-1} | x_t) of the live classes after the cond overrides -> lp[0 .. count + 1].  F64_LSE: the reference's
// float64 log-softmax (exact mode); otherwise fp32 (fast mode).
template <bool F64_LSE>
LDM_PT_HD void token_log_probs(const TokenArgs& a, const StepSchedule& s, float* lp) {
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
  // ---- constrained posterior on the live classes (constrained.py:166-197)
  const bool x_is_mask = a.tok == a.mask_id;
  // q(x_t | x_{t-1}) (l.175-185) takes three values per token: x_t's own class, any other class, [MASK]
  const float q1_same = x_is_mask ? s.lc : log_add_exp(0.0f + s.la, s.lb);
  const float q1_other = x_is_mask ? s.lc : log_add_exp(kLogEps + s.la, s.lb);
  const float q1_mask = x_is_mask ? 0.0f : kLogEps;
  const float qt_same = x_is_mask ? s.LC : log_add_exp(0.0f + s.LA, s.LB);  // q(x_t | x_0), l.166-173
  const float qt_other = x_is_mask ? s.LC : log_add_exp(kLogEps + s.LA, s.LB);
  float qmx = -INFINITY;
  for (int i = 0; i < K; ++i) {
    const int c = live_id(a, i);
    float q;
    if (c == a.mask_id) {
      q = kLogEps;  // l.189
    } else {
      float v;  // log p(x_0 = c | x_t), clamped like predict_start (base.py:140-144)
      if (F64_LSE) v = (float)(((double)a.logits[c] - (double)mx) - lse0d);
      else v = (a.logits[c] - mx) - lse0f;
      v = fminf(fmaxf(v, -70.0f), 0.0f);
      q = v - (c == a.tok ? qt_same : qt_other);  // l.188
    }
    lp[i] = q;
    qmx = fmaxf(qmx, q);
  }
  float qs = 0.f;
  for (int i = 0; i < K; ++i) qs += expf(lp[i] - qmx);
  const float lse = logf(qs) + qmx;  // torch.logsumexp
  for (int i = 0; i < K; ++i) {
    const int c = live_id(a, i);
    const float qn = lp[i] - lse;
    const float r = (c == a.mask_id) ? log_add_exp(qn + s.L1Cu, s.LCu) : log_add_exp(qn + s.LAu, s.LBu);
    const float q1 = (c == a.mask_id) ? q1_mask : (c == a.tok ? q1_same : q1_other);
    float v = fminf(fmaxf((r + q1) + lse, -70.0f), 0.0f);  // l.192-197
    // ---- constraint injection (base.py:243-284)
    if (a.strong) v = (c == a.cond_tok) ? 0.0f : kLogEps;
    else if (a.weak) v += a.weak[(long)c * a.weak_stride];
    lp[i] = v;
  }
  if (a.pad_disable) lp[a.count] = kLogEps;
}

// categorical draw over the live classes (helpers/sampling.py:81-130) -> full id.  lg = work[0 .. K) holds the
// log-probabilities on entry and is overwritten; work[K .. 3K) is scratch.
LDM_PT_HD int draw_live(const TokenArgs& a, float* lg) {
  const int K = a.count + 2;
  if (a.kind == kDeterministic) {  // first maximum in class order
    int bi = 0;
    float bv = lg[0];
    for (int i = 1; i < K; ++i)
      if (lg[i] > bv) {
        bv = lg[i];
        bi = i;
      }
    return live_id(a, bi);
  }
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const uint32_t l0 = (uint32_t)a.layout, l1 = (uint32_t)(a.layout >> 32);
  float* ex = lg + K;
  float* keep = lg + 2 * K;
  for (int i = 0; i < K; ++i) lg[i] = lg[i] / a.temperature;
  if (a.kind == kGumbel) {  // noise per class: counter word 0 = pos | (1 + c / 4) << 16, component c & 3
    for (int i = 0; i < K; ++i) {
      const int c = live_id(a, i);
      uint32_t r[4];
      philox4x32_10(a.pos | ((uint32_t)(1 + (c >> 2)) << 16), a.step, l0, l1, k0, k1, r);
      const uint32_t w = (c & 3) == 0 ? r[0] : (c & 3) == 1 ? r[1] : (c & 3) == 2 ? r[2] : r[3];
      lg[i] += -logf(-logf(u01(w) + 1e-30f) + 1e-30f);
    }
  }
  if (a.kind == kTopP || a.kind == kTopK) {
    float m1 = -INFINITY;
    for (int i = 0; i < K; ++i) m1 = fmaxf(m1, lg[i]);
    float es = 0.f;
    for (int i = 0; i < K; ++i) {
      ex[i] = expf(lg[i] - m1);
      es += ex[i];
    }
    // position in the descending (stable) order and the inclusive cumulative probability up to it
    float thr = a.kind == kTopK ? INFINITY : -INFINITY;  // top-k: the k-th largest value; top-p: no value threshold
    for (int i = 0; i < K; ++i) {
      int rank = 0;
      float cum = 0.f;
      const float li = lg[i];
      for (int o = 0; o < K; ++o) {
        const float lo = lg[o];
        const bool before = (lo > li) || (lo == li && o < i);
        if (before) {
          rank += 1;
          cum += ex[o] / es;
        } else if (o == i) {
          cum += ex[o] / es;
        }
      }
      if (a.kind == kTopP) {  // drop every class whose inclusive cumulative probability exceeds p, except the first
        keep[i] = (cum > a.top_p && rank > 0) ? 0.f : 1.f;
      } else {                // threshold = k-th largest value (sampling.py:73-78)
        keep[i] = 1.f;
        if (rank < a.top_k) thr = fminf(thr, li);
      }
    }
    for (int i = 0; i < K; ++i)
      if (keep[i] == 0.f || lg[i] < thr) lg[i] = -INFINITY;
  }
  // softmax -> inverse CDF in class order (the normaliser cancels: compare against u * total)
  float m2 = -INFINITY;
  for (int i = 0; i < K; ++i) m2 = fmaxf(m2, lg[i]);
  double base = 0.0;
  for (int i = 0; i < K; ++i) {
    ex[i] = expf(lg[i] - m2);
    base += (double)ex[i];
  }
  uint32_t r[4];
  philox4x32_10(a.pos, a.step, l0, l1, k0, k1, r);
  const double thr = (double)u01(r[0]) * base;
  double cdf = 0.0;
  int n = 0;
  for (int i = 0; i < K; ++i) {
    cdf += (double)ex[i];
    if (cdf <= thr) n += 1;
  }
  return live_id(a, n < K - 1 ? n : K - 1);
}

template <bool F64_LSE>
LDM_PT_HD int step_token(const TokenArgs& a, const StepSchedule& s, float* work) {
  token_log_probs<F64_LSE>(a, s, work);
  return draw_live(a, work);
}

}  // namespace ldm_post
