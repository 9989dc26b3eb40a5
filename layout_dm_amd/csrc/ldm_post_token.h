// The tail of one reverse step for ONE token — the single source of this arithmetic for every form it runs in:
//
//   predict_start tail   log-softmax over the C-1 non-MASK classes, clamp [-70, 0]          base.py:131-144
//   q_posterior          closed form on the token's attribute sub-vocabulary (body + [PAD] + [MASK]); every other class
//                        sits at log(1e-30)                                                  constrained.py:135-206
//   cond overrides       strong mask / refinement prior / [PAD] disable                     base.py:243-284
//   draw                 argmax | temperature, top-k, top-p, gumbel -> inverse CDF          helpers/sampling.py:81-130
//
// A token is processed by a GROUP of NL cooperating lanes; a lane owns NJ class slots.  The algorithm below is written
// once against two small policies:
//
//   G  (lane group)   lane index inside the group + the group-wide reductions (max, sum, first-argmax, inclusive prefix)
//        HostLane          NL = 1   plain C++: tests/cpu_post_token_check.cpp runs THIS code against the oracle on states
//                                   of the reference's own trajectories (tests/test_post_token_scalar.py)
//        DppGroup<16>      NL = 16  one DPP row: four tokens per wavefront (ldm_post_dpp.h) — the fused tail of the stack
//                                   kernel (kernels_stack.hip) and posterior_sample_k's default form (kernels_post.hip)
//        DppGroup<64>      NL = 64  one wavefront per token over the FULL vocabulary: the parity hooks that read or write
//                                   (B, C, S) log-probability tensors (ldm_posterior / ldm_sample_tokens)
//   M  (slot map)     SlotMap<NL, NJ, LIVE>: slot j of lane l is candidate l + NL j, in class order; LIVE = the candidates
//                     are the token's live classes only (dead classes carry 1e-30 of the mass each: leaving them out of
//                     the draw moves a CDF edge by < 1e-28 relative), else all C classes.
//
// Pure C++ apart from the DPP policy: no HIP types here.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
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

// per-token scalars, uniform over the group
struct TokenArgs {
  const float* logits;     // [n_class] logits of this token (HostLane / standalone kernels; the [MASK] column is ignored)
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

// util.py:19-21.  exp / log come from the group policy: libm on the host and in the exact numerics mode; in the fast mode
// the hardware's v_exp_f32 / v_log_f32 (~1e-6 relative, three orders below that mode's fp16 logits error)
template <class G>
LDM_PT_HD float log_add_exp(const G& g, float a, float b) {
  const float m = fmaxf(a, b);
  return m + g.log(g.exp(a - m) + g.exp(b - m));
}

LDM_PT_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                             uint32_t (&out)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
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
LDM_PT_HD bool is_live(const TokenArgs& a, int c) {
  return (c >= a.start && c < a.start + a.count) || c == a.pad_id || c == a.mask_id;
}

// ---- slot map: candidate l + NL j of lane l, in class order
template <int NL_, int NJ_, bool LIVE_>
struct SlotMap {
  static constexpr int NL = NL_, NJ = NJ_;
  static constexpr bool LIVE = LIVE_;
  int l;  // lane inside the group
  LDM_PT_HD int sidx(int j) const { return l + NL_ * j; }                          // candidate index (= scratch index)
  LDM_PT_HD int n_cand(const TokenArgs& a) const { return LIVE_ ? a.count + 2 : a.n_class; }
  LDM_PT_HD bool valid(const TokenArgs& a, int j) const { return sidx(j) < n_cand(a); }
  LDM_PT_HD int cls(const TokenArgs& a, int j) const { return LIVE_ ? live_id(a, sidx(j)) : sidx(j); }
  LDM_PT_HD bool live(const TokenArgs& a, int j) const { return valid(a, j) && (LIVE_ || is_live(a, sidx(j))); }
};

// ---- the one-lane group (host)
struct HostLane {
  static constexpr int NL = 1;
  LDM_PT_HD int lane() const { return 0; }
  LDM_PT_HD float exp(float x) const { return expf(x); }
  LDM_PT_HD float log(float x) const { return logf(x); }
  LDM_PT_HD float gmax(float v) const { return v; }
  LDM_PT_HD float gsum(float v) const { return v; }
  LDM_PT_HD double gsumd(double v) const { return v; }
  LDM_PT_HD int gsumi(int v) const { return v; }
  LDM_PT_HD void gargmax(float&, int&) const {}
  LDM_PT_HD double gscan(double v) const { return v; }  // inclusive prefix over the lanes of the group
  LDM_PT_HD void sync() const {}                        // scratch written by the group is visible to it
};

// q(x_t | x_0) and q(x_t | x_{t-1}) take three values per token: x_t's own class, any other class, [MASK]
// (constrained.py:166-185; util.py:34-40 log-one-hot of x_t)
struct QTerms {
  float qt_same, qt_other, q1_same, q1_other, q1_mask;
};
template <class G>
LDM_PT_HD QTerms q_terms(const G& g, bool x_is_mask, const StepSchedule& s) {
  QTerms k;
  k.q1_same = x_is_mask ? s.lc : log_add_exp(g, 0.0f + s.la, s.lb);
  k.q1_other = x_is_mask ? s.lc : log_add_exp(g, kLogEps + s.la, s.lb);
  k.q1_mask = x_is_mask ? 0.0f : kLogEps;
  k.qt_same = x_is_mask ? s.LC : log_add_exp(g, 0.0f + s.LA, s.LB);
  k.qt_other = x_is_mask ? s.LC : log_add_exp(g, kLogEps + s.LA, s.LB);
  return k;
}

// log p(x_0 = c | x_t) of one class from its logit and the row's (max, log-sum-exp of x - max), clamped like
// predict_start (base.py:140-144)
LDM_PT_HD float l0_f32(float x, float mx, float lse0) { return fminf(fmaxf((x - mx) - lse0, -70.0f), 0.0f); }
LDM_PT_HD float l0_f64(float x, float mx, double lse0) {
  return fminf(fmaxf((float)(((double)x - (double)mx) - lse0), -70.0f), 0.0f);
}

// ---- posterior + cond overrides: l0[j] = log p(x_0 = cls(j) | x_t) of the lane's slots (ignored for [MASK] and for
// dead / invalid slots) -> lp[j] = log p(x_{t-1} = cls(j) | x_t) after the overrides; -inf for invalid slots.
// k = q_terms(g, a.tok == a.mask_id, s): the same for every token of an attribute that is / is not [MASK], so a caller
// with many tokens per step computes the ten variants once
template <class G, class M>
LDM_PT_HD void token_log_probs(const G& g, const M& m, const TokenArgs& a, const StepSchedule& s, const QTerms& kq,
                               const float (&l0)[M::NJ], float (&lp)[M::NJ]) {
  constexpr int NJ = M::NJ;
  // (scalars: selecting between the MEMBERS of a struct per class makes hipcc keep the struct in private memory and
  //  index it)
  const float qt_same = kq.qt_same, qt_other = kq.qt_other, q1_same = kq.q1_same, q1_other = kq.q1_other, q1_mask = kq.q1_mask;
  float q[NJ];
  float qmx = -INFINITY;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < NJ; ++j) {
    q[j] = -INFINITY;
    if (m.live(a, j)) {
      const int c = m.cls(a, j);
      q[j] = (c == a.mask_id) ? kLogEps                                            // constrained.py:189
                              : l0[j] - (c == a.tok ? qt_same : qt_other);         // l.188
      qmx = fmaxf(qmx, q[j]);
    }
  }
  qmx = g.gmax(qmx);
  float qs = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < NJ; ++j)
    if (m.live(a, j)) qs += g.exp(q[j] - qmx);
  qs = g.gsum(qs);
  const float lse = g.log(qs) + qmx;  // torch.logsumexp
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < NJ; ++j) {
    const int c = m.cls(a, j);
    float v;
    if (m.live(a, j)) {
      const float qn = q[j] - lse;
      const float r = (c == a.mask_id) ? log_add_exp(g, qn + s.L1Cu, s.LCu) : log_add_exp(g, qn + s.LAu, s.LBu);
      const float q1 = (c == a.mask_id) ? q1_mask : (c == a.tok ? q1_same : q1_other);
      v = fminf(fmaxf((r + q1) + lse, -70.0f), 0.0f);  // l.192-197
    } else {
      v = kLogEps;  // p_to_f_log fill (layout_tokenizer.py:544)
    }
    if (m.valid(a, j)) {
      // ---- constraint injection (base.py:243-284)
      if (a.strong) v = (c == a.cond_tok) ? 0.0f : kLogEps;
      else if (a.weak) v += a.weak[(long)c * a.weak_stride];
      if (a.pad_disable && c == a.pad_id) v = kLogEps;
    } else {
      v = -INFINITY;
    }
    lp[j] = v;
  }
}

template <class G, class M>
LDM_PT_HD void token_log_probs(const G& g, const M& m, const TokenArgs& a, const StepSchedule& s,
                               const float (&l0)[M::NJ], float (&lp)[M::NJ]) {
  token_log_probs(g, m, a, s, q_terms(g, a.tok == a.mask_id, s), l0, lp);
}

// only the [PAD] disabling (the stage between ldm_relation_update and the draw: base.py:272-284)
template <class M>
LDM_PT_HD void pad_disable_only(const M& m, const TokenArgs& a, float (&lp)[M::NJ]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < M::NJ; ++j)
    if (a.pad_disable && m.valid(a, j) && m.cls(a, j) == a.pad_id) lp[j] = kLogEps;
}

struct Draw {
  int token;   // full id
  float gap;   // deterministic: log-probability gap between the winner and the runner-up (+inf for the stochastic kinds)
};

// A strong-masked position (base.py:245-251) carries log-probability 0 on its conditioned token and log(1e-30) on every
// other class: argmax returns the conditioned token, and so does every stochastic draw this sampler can make — the other
// classes hold < 2e-28 of the mass, below the 2^-24 resolution of the uniform (and gumbel noise spans < 20 of the 69 nats
// between them).  Callers skip the posterior and the draw for such tokens; the tokens are identical either way
// (tests/test_hip_parity.py::test_strong_mask_shortcut_is_an_identity).
LDM_PT_HD bool strong_shortcut(const TokenArgs& a) { return a.strong && a.cond_tok >= 0 && a.cond_tok < a.n_class; }

// ---- top-k / top-p: walk the candidates o in class order; for the class of each slot j accumulate the probability
// (and, RANK, the number) of the candidates that precede it in the descending stable order, itself included.
// Branch-free on purpose: the 16 lanes of a group rarely agree, and selects cost a third of the divergent form.
template <bool RANK, class M>
LDM_PT_HD void order_walk(const M& m, const TokenArgs& a, const float (&lg)[M::NJ], const float* sc_lg,
                          const float* sc_pr, int n_walk, bool cand_all, float (&cum)[M::NJ], int (&rank)[M::NJ]) {
  constexpr int NJ = M::NJ;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < NJ; ++j) {
    cum[j] = 0.f;
    rank[j] = -1;  // the class itself is counted below (slots without a class are never read)
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 4  // (four candidates' scratch reads in flight: with one wavefront per SIMD nothing else hides their latency)
#endif
  for (int oi = 0; oi < n_walk; ++oi) {
    const int o = (M::LIVE || cand_all) ? oi : live_id(a, oi);  // candidate (= scratch) index, increasing with class
    const float ol = sc_lg[o];
    const float op = sc_pr[o];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < NJ; ++j) {
      const bool upto = (ol > lg[j]) | ((ol == lg[j]) & (o <= m.sidx(j)));  // o precedes slot j's class, or is it
      cum[j] += upto ? op : 0.f;
      if (RANK) rank[j] += upto ? 1 : 0;
    }
  }
}

// ---- categorical draw over the group's candidates (helpers/sampling.py:81-130).  lp is consumed.
// sc_lg / sc_pr: scratch of n_cand floats each, private to the group (top-k / top-p only), indexed by candidate.
// cand_all: the rank loop of top-k / top-p walks every candidate; else (full-vocabulary map on a posterior it computed
// itself) only the live ones, which alone can carry mass.
template <class G, class M>
LDM_PT_HD Draw draw_token(const G& g, const M& m, const TokenArgs& a, float (&lp)[M::NJ], float* sc_lg, float* sc_pr,
                          bool cand_all) {
  constexpr int NJ = M::NJ;
  Draw out;
  out.gap = INFINITY;
  if (a.kind == kDeterministic) {  // first maximum in class order
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < NJ; ++j)
      if (m.valid(a, j) && lp[j] > bv) {
        bv = lp[j];
        bi = m.cls(a, j);
      }
    g.gargmax(bv, bi);
    float second = -INFINITY;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < NJ; ++j)
      if (m.valid(a, j) && m.cls(a, j) != bi) second = fmaxf(second, lp[j]);
    second = g.gmax(second);
    out.token = bi;
    out.gap = bv - second;
    return out;
  }
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const uint32_t y0 = (uint32_t)a.layout, y1 = (uint32_t)(a.layout >> 32);
  float lg[NJ];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < NJ; ++j) lg[j] = m.valid(a, j) ? lp[j] / a.temperature : -INFINITY;
  if (a.kind == kGumbel) {  // noise per class: counter word 0 = pos | (1 + c / 4) << 16, component c & 3
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < NJ; ++j)
      if (m.valid(a, j)) {
        const int c = m.cls(a, j);
        uint32_t r[4];
        philox4x32_10(a.pos | ((uint32_t)(1 + (c >> 2)) << 16), a.step, y0, y1, k0, k1, r);
        const uint32_t w = (c & 3) == 0 ? r[0] : (c & 3) == 1 ? r[1] : (c & 3) == 2 ? r[2] : r[3];
        lg[j] += -g.log(-g.log(u01(w) + 1e-30f) + 1e-30f);
      }
  }
  if (a.kind == kTopP || a.kind == kTopK) {
    // softmax of lg (top-p's cumulative probabilities)
    float m1 = -INFINITY;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < NJ; ++j) m1 = fmaxf(m1, lg[j]);
    m1 = g.gmax(m1);
    float ex[NJ], es = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < NJ; ++j) {
      ex[j] = m.valid(a, j) ? g.exp(lg[j] - m1) : 0.f;
      es += ex[j];
    }
    es = g.gsum(es);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < NJ; ++j)
      if (m.valid(a, j)) {
        sc_lg[m.sidx(j)] = lg[j];
        sc_pr[m.sidx(j)] = ex[j] / es;
      }
    g.sync();
    // inclusive cumulative probability of every class in the descending (stable) order, and for top-k its position
    float cum[NJ];
    int rank[NJ];
    const int n_walk = (M::LIVE || cand_all) ? m.n_cand(a) : a.count + 2;
    if (a.kind == kTopP) {  // drop every class whose inclusive cumulative probability exceeds p, except the first
      order_walk<false>(m, a, lg, sc_lg, sc_pr, n_walk, cand_all, cum, rank);
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = 0; j < NJ; ++j)
        if (m.valid(a, j) && lg[j] > bv) {
          bv = lg[j];
          bi = m.sidx(j);
        }
      g.gargmax(bv, bi);  // first maximum in class order = position 0
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = 0; j < NJ; ++j)
        if (cum[j] > a.top_p && m.sidx(j) != bi) lg[j] = -INFINITY;
    } else {                // threshold = k-th largest value (sampling.py:73-78)
      order_walk<true>(m, a, lg, sc_lg, sc_pr, n_walk, cand_all, cum, rank);
      float thr = INFINITY;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = 0; j < NJ; ++j)
        if (m.valid(a, j) && rank[j] < a.top_k) thr = fminf(thr, lg[j]);
      thr = -g.gmax(-thr);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = 0; j < NJ; ++j)
        if (lg[j] < thr) lg[j] = -INFINITY;
    }
  }
  // softmax -> inverse CDF in class order (the normaliser cancels: compare against u * total)
  float m2 = -INFINITY;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < NJ; ++j) m2 = fmaxf(m2, lg[j]);
  m2 = g.gmax(m2);
  double cdf[NJ];
  double base = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < NJ; ++j) {  // slot j of every lane precedes slot j + 1 of every lane
    const double pr = m.valid(a, j) ? (double)g.exp(lg[j] - m2) : 0.0;
    cdf[j] = base + g.gscan(pr);
    base += g.gsumd(pr);
  }
  uint32_t r[4];
  philox4x32_10(a.pos, a.step, y0, y1, k0, k1, r);
  const double thr = (double)u01(r[0]) * base;
  int n = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < NJ; ++j)
    if (m.valid(a, j) && cdf[j] <= thr) n += 1;
  n = g.gsumi(n);
  const int last = m.n_cand(a) - 1;
  n = n < last ? n : last;
  out.token = M::LIVE ? live_id(a, n) : n;
  return out;
}

// ---- the one-lane form: log-softmax, posterior, overrides, draw on the token's live classes.  work: 2 * (count + 2)
// floats of scratch (top-k / top-p).  F64_LSE: the reference's float64 log-softmax (exact mode); otherwise fp32 (fast).
constexpr int kHostMaxLive = 64;
template <bool F64_LSE>
LDM_PT_HD Draw step_token_draw(const TokenArgs& a, const StepSchedule& s, float* work) {
  const int C = a.n_class, K = a.count + 2;
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
  const HostLane g;
  const SlotMap<1, kHostMaxLive, true> m{0};
  float l0[kHostMaxLive], lp[kHostMaxLive];
  for (int j = 0; j < kHostMaxLive; ++j) {
    l0[j] = 0.f;
    if (j < K - 1) l0[j] = F64_LSE ? l0_f64(a.logits[live_id(a, j)], mx, lse0d) : l0_f32(a.logits[live_id(a, j)], mx, lse0f);
  }
  token_log_probs(g, m, a, s, l0, lp);
  return draw_token(g, m, a, lp, work, work + K, true);
}
template <bool F64_LSE>
LDM_PT_HD int step_token(const TokenArgs& a, const StepSchedule& s, float* work) {
  return step_token_draw<F64_LSE>(a, s, work).token;
}

}  // namespace ldm_post
