// Fused tail of one reverse step, one 64-lane wavefront per token (lane l owns classes l, l+64,
// l+128 of the <=192-class vocabulary), everything in registers + wave shuffles:
//
//   predict_start tail   log_softmax (float64) over the C-1 non-MASK classes, -70 for MASK,
//                        clamp [-70,0]                 categorical_diffusion/base.py:131-144
//   q_posterior          per-attribute sub-vocabulary posterior  constrained.py:135-206
//                        (q_pred 112-133, q_pred_one_timestep 92-110, log helpers util.py:15-27,
//                         Converter gather/scatter helpers/layout_tokenizer.py:540-557)
//   cond overrides       strong mask / refinement prior / PAD disable   base.py:243-284
//   sample               argmax | temperature, top-k, top-p, gumbel -> softmax -> multinomial
//                        helpers/sampling.py:81-130
//
// The reference materialises (B,C,S) log-one-hot / log-prob tensors around ~170 tiny ATen launches
// per step; here the state stays int32 tokens and the only HBM traffic is the
// logits row in and one token out (plus the optional (B,C,S) parity dump).
// torch.multinomial's stream is replaced by counter-based Philox4x32-10 keyed by
// (seed, global layout index, reverse-step index, position) => results do not depend on how the
// batch is split over calls or GPUs.
#include "ldm_kernels.h"

namespace ldm {

// ---- wave-wide reductions WITHOUT the LDS: a __shfl_xor is a ds_bpermute_b32 (an LDS round trip, ~130 cycles, and the
// 6 steps of a reduction are a dependent chain) and this kernel executed ~70 of them per token — its waves lived ~22k
// cycles for ~2k cycles of arithmetic.  Butterfly on DPP moves inside a 16-lane row (quad_perm, row_half_mirror,
// row_mirror: one VALU issue each) and the two cross-row exchanges through gfx950's v_permlane16_swap / v_permlane32_swap.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;
// values of the other half of every exchange step, for a 32-bit payload
#define LDM_BFLY32(v, COMBINE)                                                                              \
  do {                                                                                                      \
    { const int o = dpp_mov<kDppXor1>(v); COMBINE(o); }                                                      \
    { const int o = dpp_mov<kDppXor2>(v); COMBINE(o); }                                                      \
    { const int o = dpp_mov<kDppHalfMirror>(v); COMBINE(o); }                                                \
    { const int o = dpp_mov<kDppMirror>(v); COMBINE(o); }                                                    \
  } while (0)
__device__ __forceinline__ float wmax(float v) {
  int x = __float_as_int(v);
#define LDM_CMB(o) x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(o)))
  LDM_BFLY32(x, LDM_CMB);
  { const auto s = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false); x = __float_as_int(fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]))); }
  { const auto s = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false); x = __float_as_int(fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]))); }
#undef LDM_CMB
  return __int_as_float(x);
}
__device__ __forceinline__ float wsum(float v) {
  int x = __float_as_int(v);
#define LDM_CMB(o) x = __float_as_int(__int_as_float(x) + __int_as_float(o))
  LDM_BFLY32(x, LDM_CMB);
  { const auto s = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false); x = __float_as_int(__uint_as_float(s[0]) + __uint_as_float(s[1])); }
  { const auto s = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false); x = __float_as_int(__uint_as_float(s[0]) + __uint_as_float(s[1])); }
#undef LDM_CMB
  return __int_as_float(x);
}
__device__ __forceinline__ int wsumi(int x) {
#define LDM_CMB(o) x = x + (o)
  LDM_BFLY32(x, LDM_CMB);
  { const auto s = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false); x = (int)(s[0] + s[1]); }
  { const auto s = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false); x = (int)(s[0] + s[1]); }
#undef LDM_CMB
  return x;
}
template <int CTRL>
__device__ __forceinline__ double dpp_movd(double v) {
  return __hiloint2double(dpp_mov<CTRL>(__double2hiint(v)), dpp_mov<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ double wsumd(double v) {
  v += dpp_movd<kDppXor1>(v);
  v += dpp_movd<kDppXor2>(v);
  v += dpp_movd<kDppHalfMirror>(v);
  v += dpp_movd<kDppMirror>(v);
  {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
  {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
  return v;
}
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside a row (row_shr 1 / 2 / 4 / 8, zeros shifted in), then the
// totals of the preceding rows through row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_movd_rows(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ double wscan(double v, int lane) {
  (void)lane;
  v += dpp_movd_rows<0x111, 0xF>(v);  // row_shr:1
  v += dpp_movd_rows<0x112, 0xF>(v);  // row_shr:2
  v += dpp_movd_rows<0x114, 0xF>(v);  // row_shr:4
  v += dpp_movd_rows<0x118, 0xF>(v);  // row_shr:8
  v += dpp_movd_rows<0x142, 0xA>(v);  // row_bcast:15 -> rows 1, 3
  v += dpp_movd_rows<0x143, 0xC>(v);  // row_bcast:31 -> rows 2, 3
  return v;
}
// broadcast of lane 63 (the total of an inclusive scan)
__device__ __forceinline__ double wlast(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// util.py:19-21
__device__ __forceinline__ float log_add_exp(float a, float b) {
  const float m = fmaxf(a, b);
  return m + logf(expf(a - m) + expf(b - m));
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
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
__device__ __forceinline__ float u01(uint32_t x) {  // strictly inside (0,1), exact in fp32
  return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f;  // 2^-23
}

constexpr int NJ = 3;  // classes per lane (C <= 192)

__global__ __launch_bounds__(256) void posterior_sample_k(PostArgs p) {
  __shared__ float sh_lg[4][192];
  __shared__ float sh_pr[4][192];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  const int M = p.B * p.S;
  if (row >= M) return;  // whole wave exits together; no block-level barrier is used below
  const int b = row / p.S, s = row % p.S;
  const int C = p.v.n_class;
  const int attr = s % p.v.n_attr;
  const int pad_id = p.v.pad_id, mask_id = p.v.mask_id;

  float lp[NJ];  // full-vocabulary log p(x_{t-1} | x_t) for this lane's classes
  if (p.logp_in) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      lp[j] = (c < C) ? p.logp_in[((size_t)b * C + c) * p.S + s] : -INFINITY;
    }
  } else {
    // ---- log p(x0 | xt): float64 log-softmax over classes [0, C-1)
    const float* lrow = p.logits + (size_t)row * p.ldl;
    float xv[NJ];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      xv[j] = (c < C - 1) ? lrow[c] : -INFINITY;
      mx = fmaxf(mx, xv[j]);
    }
    mx = wmax(mx);
    float l0[NJ];
    if (p.f32_lse) {  // fast numerics mode: ~1e-7 relative, far inside its 1e-3 logits budget
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        if (c < C - 1) se += expf(xv[j] - mx);
      }
      const float lse0 = logf(wsum(se));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        const float v = (c < C - 1) ? (xv[j] - mx) - lse0 : -70.0f;
        l0[j] = fminf(fmaxf(v, -70.0f), 0.0f);
      }
    } else {
      double se = 0.0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        if (c < C - 1) se += exp((double)xv[j] - (double)mx);
      }
      se = wsumd(se);
      const double lse0 = log(se);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        float v = (c < C - 1) ? (float)(((double)xv[j] - (double)mx) - lse0) : -70.0f;
        l0[j] = fminf(fmaxf(v, -70.0f), 0.0f);
      }
    }
    // ---- constrained posterior in this token's attribute sub-vocabulary
    const int T1 = p.T + 1;
    const int t = p.t_post;
    const int u = (t - 1 + T1) % T1;  // constrained.py:114
    auto sch = [&](int kind, int idx) { return p.sched[((size_t)kind * p.v.n_attr + attr) * T1 + idx]; };
    const float la = sch(kLogAt, t), lb = sch(kLogBt, t), lc = sch(kLogCt, t);
    const float LA = sch(kLogCumAt, t), LB = sch(kLogCumBt, t), LC = sch(kLogCumCt, t);
    const float LAu = sch(kLogCumAt, u), LBu = sch(kLogCumBt, u), LCu = sch(kLogCumCt, u);
    const float L1Cu = sch(kLog1mCumCt, u);
    const int tok = p.tokens[row];
    const bool x_is_mask = (tok == mask_id);
    const int start = p.v.start[attr], cnt = p.v.count[attr];

    bool live[NJ];
    float q[NJ], q1[NJ];
    float qmx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      live[j] = (c < C) && ((c >= start && c < start + cnt) || c == pad_id || c == mask_id);
      q[j] = -INFINITY;
      q1[j] = 0.f;
      if (live[j]) {
        if (c == mask_id) {
          q[j] = kLogEps;                       // constrained.py:189
          q1[j] = x_is_mask ? 0.0f : kLogEps;   // l.179-185
        } else {
          float qt;
          if (x_is_mask) {
            qt = LC;   // l.169-173
            q1[j] = lc;
          } else {
            const float e = (c == tok) ? 0.0f : kLogEps;  // log-one-hot of x_t (util.py:34-40)
            qt = log_add_exp(e + LA, LB);
            q1[j] = log_add_exp(e + la, lb);
          }
          q[j] = l0[j] - qt;  // l.188
        }
        qmx = fmaxf(qmx, q[j]);
      }
    }
    qmx = wmax(qmx);
    float qs = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (live[j]) qs += expf(q[j] - qmx);
    qs = wsum(qs);
    const float lse = logf(qs) + qmx;  // torch.logsumexp
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (live[j]) {
        const float qn = q[j] - lse;
        const float r = (c == mask_id) ? log_add_exp(qn + L1Cu, LCu) : log_add_exp(qn + LAu, LBu);
        lp[j] = fminf(fmaxf((r + q1[j]) + lse, -70.0f), 0.0f);  // l.192-197
      } else {
        lp[j] = (c < C) ? kLogEps : -INFINITY;  // p_to_f_log fill
      }
    }
    // ---- constraint injection (base.py:243-284)
    const int cs = p.cond_seq ? p.cond_seq[row] : -1;
    const bool strong = p.strong && p.strong[row];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c >= C) continue;
      if (strong) lp[j] = (c == cs) ? 0.0f : kLogEps;
      else if (p.weak) lp[j] += p.weak[((size_t)b * C + c) * p.S + s];
    }
  }
  // disable [PAD] where the number of elements is known (base.py:272-284).  Also on the logp_in path: for
  // cond=relation the reference applies it AFTER the logit adjustment, i.e. between ldm_relation_update and the draw.
  if (p.pad_disable && p.cond_seq && attr != 0 && p.cond_seq[row] != pad_id) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (lane + 64 * j == pad_id) lp[j] = kLogEps;
  }
  if (p.logp_out) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < C) p.logp_out[((size_t)b * C + c) * p.S + s] = lp[j];
    }
  }
  if (!p.tokens_out) return;

  // ---- categorical draw (helpers/sampling.py:81-130)
  int result;
  if (p.kind == 0) {  // deterministic: first maximum
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < C && lp[j] > bv) { bv = lp[j]; bi = c; }
    }
    {
      auto take = [&](float ov, int oi) {
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      };
      take(__int_as_float(dpp_mov<kDppXor1>(__float_as_int(bv))), dpp_mov<kDppXor1>(bi));
      take(__int_as_float(dpp_mov<kDppXor2>(__float_as_int(bv))), dpp_mov<kDppXor2>(bi));
      take(__int_as_float(dpp_mov<kDppHalfMirror>(__float_as_int(bv))), dpp_mov<kDppHalfMirror>(bi));
      take(__int_as_float(dpp_mov<kDppMirror>(__float_as_int(bv))), dpp_mov<kDppMirror>(bi));
      {
        const auto sv = __builtin_amdgcn_permlane16_swap((unsigned)__float_as_int(bv), (unsigned)__float_as_int(bv), false, false);
        const auto si = __builtin_amdgcn_permlane16_swap((unsigned)bi, (unsigned)bi, false, false);
        bv = __uint_as_float(sv[0]); bi = (int)si[0];
        take(__uint_as_float(sv[1]), (int)si[1]);
      }
      {
        const auto sv = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(bv), (unsigned)__float_as_int(bv), false, false);
        const auto si = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
        bv = __uint_as_float(sv[0]); bi = (int)si[0];
        take(__uint_as_float(sv[1]), (int)si[1]);
      }
    }
    result = bi;
  } else {
    const uint64_t seed = p.rng[0];
    const uint64_t layout = p.rng[1] + (uint64_t)p.layout_off + (uint64_t)b;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    float lg[NJ];
    const float inv_t = 1.0f / p.temperature;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      lg[j] = (c < C) ? lp[j] / p.temperature : -INFINITY;
    }
    (void)inv_t;
    if (p.kind == 4) {  // gumbel noise per class: counter word 0 = pos | (1 + c/4) << 16
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        if (c < C) {
          uint32_t r[4];
          philox4x32_10((uint32_t)s | ((uint32_t)(1 + (c >> 2)) << 16), (uint32_t)p.step, (uint32_t)layout,
                        (uint32_t)(layout >> 32), k0, k1, r);
          const float uu = u01(r[c & 3]);
          lg[j] += -logf(-logf(uu + 1e-30f) + 1e-30f);
        }
      }
    }
    if (p.kind == 2 || p.kind == 3) {
      // softmax of lg (needed for top-p's cumulative probabilities)
      float m1 = wmax(fmaxf(fmaxf(lg[0], lg[1]), lg[2]));
      float ex[NJ], es = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        ex[j] = (c < C) ? expf(lg[j] - m1) : 0.f;
        es += ex[j];
      }
      es = wsum(es);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        if (c < C) {
          sh_lg[wave][c] = lg[j];
          sh_pr[wave][c] = ex[j] / es;
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave are done
      __builtin_amdgcn_wave_barrier();
      float cum[NJ] = {0.f, 0.f, 0.f};
      int rank[NJ] = {0, 0, 0};
      // Only classes of this token's sub-vocabulary can carry mass (every other class sits at
      // log(1e-30), layout_tokenizer.py:544): walk body + PAD + MASK instead of all C classes.  With a
      // full-vocabulary logp_in (ldm_sample_tokens hook) every class is a candidate.
      const int a_start = p.v.start[attr], a_cnt = p.v.count[attr];
      const int n_cand = p.logp_in ? C : a_cnt + 2;
      for (int oi = 0; oi < n_cand; ++oi) {
        const int o = p.logp_in ? oi : (oi < a_cnt ? a_start + oi : (oi == a_cnt ? pad_id : mask_id));
        const float ol = sh_lg[wave][o];
        const float op = sh_pr[wave][o];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = lane + 64 * j;
          const bool before = (ol > lg[j]) || (ol == lg[j] && o < c);  // sorted-descending position
          if (before) { rank[j] += 1; cum[j] += op; }
          else if (o == c) cum[j] += op;  // inclusive
        }
      }
      if (p.kind == 2) {  // top-p: drop every class whose inclusive cumulative prob exceeds p (rank>0)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (cum[j] > p.top_p && rank[j] > 0) lg[j] = -INFINITY;
      } else {  // top-k: threshold = k-th largest value (sampling.py:73-78)
        float thr = INFINITY;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = lane + 64 * j;
          if (c < C && rank[j] < p.top_k) thr = fminf(thr, lg[j]);
        }
        thr = -wmax(-thr);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (lg[j] < thr) lg[j] = -INFINITY;
      }
    }
    // softmax -> inverse-CDF draw in class order
    const float m2 = wmax(fmaxf(fmaxf(lg[0], lg[1]), lg[2]));
    double pr[NJ], tot[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      pr[j] = (c < C) ? (double)expf(lg[j] - m2) : 0.0;
    }
    double base = 0.0;
    double cdf[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const double sc = wscan(pr[j], lane);
      tot[j] = wlast(sc);
      cdf[j] = base + sc;
      base += tot[j];
    }
    uint32_t r[4];
    philox4x32_10((uint32_t)s, (uint32_t)p.step, (uint32_t)layout, (uint32_t)(layout >> 32), k0, k1, r);
    const double thr = (double)u01(r[0]) * base;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < C && cdf[j] <= thr) cnt += 1;
    }
    cnt = wsumi(cnt);
    result = cnt < C - 1 ? cnt : C - 1;
  }
  if (lane == 0) p.tokens_out[row] = result;
  if (p.x_next) {
    // the wave that drew the token also writes the row the next reverse step starts from (the separate embedding
    // launch cannot overlap anything: the stack kernel's workgroups own whole CUs)
    const int tok = __builtin_amdgcn_readfirstlane(result);
    const float4* e = reinterpret_cast<const float4*>(p.emb + (size_t)tok * p.D);
    const float4* ps = reinterpret_cast<const float4*>(p.pos + (size_t)s * p.D);
    float4* o = reinterpret_cast<float4*>(p.x_next + (size_t)row * p.ldx);
    const int nvec = p.D >> 2;
    for (int c = lane; c < nvec; c += 64) {
      const float4 x = e[c], y = ps[c];
      o[c] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
  }
}

__global__ void set_rng_k(uint64_t* rng, uint64_t seed, uint64_t first_layout) {
  rng[0] = seed;
  rng[1] = first_layout;
}
void launch_set_rng(uint64_t* rng, uint64_t seed, uint64_t first_layout, hipStream_t st) {
  hipLaunchKernelGGL(set_rng_k, dim3(1), dim3(1), 0, st, rng, seed, first_layout);
}

void launch_posterior_sample(const PostArgs& p, hipStream_t st) {
  const int M = p.B * p.S;
  hipLaunchKernelGGL(posterior_sample_k, dim3((M + 3) / 4), dim3(256), 0, st, p);
}

}  // namespace ldm
