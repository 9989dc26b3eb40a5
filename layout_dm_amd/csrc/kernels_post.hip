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
#include <cstdlib>

#include "ldm_kernels.h"
#include "ldm_post_dpp.h"

namespace ldm {

// The arithmetic of the tail lives in ldm_post_token.h (one source for this kernel, the fused tail of the stack kernel
// and the host check); this kernel supplies the lane groups and the memory traffic.
//
//   NL = 16, LIVE   (default)  one DPP row per token, four tokens per wavefront, slots = the token's live classes only
//                              (<= 48): the step path (tokens in, tokens out) of every mode that does not run the tail
//                              inside the stack kernel — exact / split numerics, cond=relation's two half-steps, geometries
//                              off the layout-resident kernels
//   NL = 64, full vocabulary   one wavefront per token: the parity hooks that read or write (B, C, S) tensors
//                              (ldm_posterior, ldm_sample_tokens), and vocabularies whose live set exceeds 48 (vanilla)
template <int NL, bool LIVE, bool FAST>
__global__ __launch_bounds__(256) void posterior_sample_k(PostArgs p) {
  constexpr int GPW = 64 / NL;            // tokens per wavefront
  constexpr int NJ = LIVE ? 3 : 192 / NL; // class slots per lane
  constexpr int NSC = LIVE ? 48 : 192;
  static_assert(LIVE || NL == 64, "full-vocabulary map: one wavefront per token");
  __shared__ float sh_lg[4 * GPW][NSC];
  __shared__ float sh_pr[4 * GPW][NSC];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = lane / NL;
  const int row = (blockIdx.x * 4 + wave) * GPW + grp;
  const int M = p.B * p.S;
  if (row >= M) return;  // whole groups (DPP rows / wavefronts) leave together; no block-level barrier is used below
  const int b = row / p.S, s = row % p.S;
  const int C = p.v.n_class;
  const int attr = s % p.v.n_attr;
  const ldm_post::DppGroup<NL, FAST> g{lane % NL};
  const ldm_post::SlotMap<NL, NJ, LIVE> m{lane % NL};

  ldm_post::TokenArgs a{};
  a.tok = p.tokens ? p.tokens[row] : -1;
  a.start = p.v.start[attr];
  a.count = p.v.count[attr];
  a.pad_id = p.v.pad_id;
  a.mask_id = p.v.mask_id;
  a.n_class = C;
  a.cond_tok = p.cond_seq ? p.cond_seq[row] : -1;
  a.strong = p.strong && p.strong[row];
  a.weak = p.weak ? p.weak + (size_t)b * C * p.S + s : nullptr;  // (B, C, S)
  a.weak_stride = p.S;
  a.pad_disable = p.pad_disable && p.cond_seq && attr != 0 && a.cond_tok != p.v.pad_id;  // base.py:272-284
  a.kind = p.kind;
  a.temperature = p.temperature;
  a.top_p = p.top_p;
  a.top_k = p.top_k;
  a.pos = (uint32_t)s;
  a.step = (uint32_t)p.step;
  if (p.tokens_out && p.kind != 0) {
    a.layout = p.rng[1] + (uint64_t)p.layout_off + (uint64_t)b;
    a.seed = p.rng[0];
  }

  if (LIVE && ldm_post::strong_shortcut(a)) {  // conditioned token: every sampler returns it (ldm_post_token.h)
    if (g.lane() == 0) p.tokens_out[row] = a.cond_tok;
    if (p.x_next) {
      const float4* e = reinterpret_cast<const float4*>(p.emb + (size_t)a.cond_tok * p.D);
      const float4* ps = reinterpret_cast<const float4*>(p.pos + (size_t)s * p.D);
      float4* o = reinterpret_cast<float4*>(p.x_next + (size_t)row * p.ldx);
      for (int c = g.lane(); c < (p.D >> 2); c += NL) {
        const float4 x = e[c], y = ps[c];
        o[c] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
      }
    }
    return;
  }
  float lp[NJ];  // log p(x_{t-1} | x_t) of this lane's slots
  float absmax = 0.f;
  if (p.logp_in) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      lp[j] = m.valid(a, j) ? p.logp_in[p.logp_tm ? ((size_t)b * p.S + s) * C + m.cls(a, j) : ((size_t)b * C + m.cls(a, j)) * p.S + s]
                            : -INFINITY;
    // disable [PAD] where the number of elements is known: for cond=relation the reference applies it AFTER the logit
    // adjustment, i.e. between ldm_relation_update and the draw
    ldm_post::pad_disable_only(m, a, lp);
  } else {
    // ---- log p(x0 | xt): log-softmax over classes [0, C-1) (float64 like base.py:137, or fp32 in the fast mode)
    const float* lrow = p.logits + (size_t)row * p.ldl;
    // the lane's classes c = lane, lane + NL, ... are loaded ONCE, into registers, with every load issued before the first use (r06:
    // the three passes below used to re-read the row through a rolled loop — one dependent global load per iteration, ~10 exposed
    // L2 round trips per lane in front of the f64 exponentials: 111 us per launch in the split mode's profile, more than the
    // vocabulary head it follows).  Same values, same summation order: results are bit-identical.
    constexpr int NX = (192 + NL - 1) / NL;   // ldm_create: at most 192 classes
    float xv[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int c = g.lane() + i * NL;
      xv[i] = c < C - 1 ? lrow[c] : -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      mx = fmaxf(mx, xv[i]);
      absmax = fmaxf(absmax, g.lane() + i * NL < C - 1 ? fabsf(xv[i]) : 0.f);
    }
    mx = g.gmax(mx);
    absmax = g.gmax(absmax);
    float l0[NJ];
    if (FAST) {  // fast numerics mode (p.f32_lse): ~1e-6 relative, far inside its 1e-3 logits budget
      float se = 0.f;
#pragma unroll
      for (int i = 0; i < NX; ++i) se += g.exp(xv[i] - mx);   // (classes beyond the row: exp(-inf) = 0)
      const float lse0 = g.log(g.gsum(se));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = m.cls(a, j);
        l0[j] = (m.valid(a, j) && c < C - 1) ? ldm_post::l0_f32(lrow[c], mx, lse0) : -70.0f;
      }
    } else {
      double se = 0.0;
#pragma unroll
      for (int i = 0; i < NX; ++i) se += exp((double)xv[i] - (double)mx);   // (classes beyond the row: exp(-inf) = 0)
      const double lse0 = log(g.gsumd(se));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = m.cls(a, j);
        l0[j] = (m.valid(a, j) && c < C - 1) ? ldm_post::l0_f64(lrow[c], mx, lse0) : -70.0f;
      }
    }
    const int T1 = p.T + 1;
    const int t = p.t_post;
    const int u = (t - 1 + T1) % T1;  // constrained.py:114
    auto sch = [&](int kind, int idx) { return p.sched[((size_t)kind * p.v.n_attr + attr) * T1 + idx]; };
    const ldm_post::StepSchedule sc{sch(kLogAt, t),    sch(kLogBt, t),    sch(kLogCt, t),    sch(kLogCumAt, t),
                                    sch(kLogCumBt, t), sch(kLogCumCt, t), sch(kLogCumAt, u), sch(kLogCumBt, u),
                                    sch(kLogCumCt, u), sch(kLog1mCumCt, u)};
    ldm_post::token_log_probs(g, m, a, sc, l0, lp);
  }
  if (p.logp_out) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (m.valid(a, j))
        p.logp_out[p.logp_tm ? ((size_t)b * p.S + s) * C + m.cls(a, j) : ((size_t)b * C + m.cls(a, j)) * p.S + s] = lp[j];
  }
  if (!p.tokens_out) return;

  // ---- categorical draw (helpers/sampling.py:81-130)
  const ldm_post::Draw d = ldm_post::draw_token(g, m, a, lp, sh_lg[wave * GPW + grp], sh_pr[wave * GPW + grp],
                                                p.logp_in != nullptr);
  if (g.lane() == 0) {
    p.tokens_out[row] = d.token;
    // near-tie report (deterministic decoding): the winner's lead over the runner-up is inside what the mode's logits
    // error can move — the caller re-decides this layout in the exact mode (ldm_tie_flags_*)
    if (p.tie_flags && !p.logp_in && d.gap < fmaxf(p.tie_rel * absmax, p.tie_abs)) p.tie_flags[b] = 1;
  }
  if (p.x_next) {
    // the group that drew the token also writes the row the next reverse step starts from (a separate embedding
    // launch cannot overlap anything: the stack kernel's workgroups own whole CUs)
    const float4* e = reinterpret_cast<const float4*>(p.emb + (size_t)d.token * p.D);
    const float4* ps = reinterpret_cast<const float4*>(p.pos + (size_t)s * p.D);
    float4* o = reinterpret_cast<float4*>(p.x_next + (size_t)row * p.ldx);
    const int nvec = p.D >> 2;
    for (int c = g.lane(); c < nvec; c += NL) {
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
  int live_max = 0;
  for (int a = 0; a < p.v.n_attr; ++a) live_max = live_max > p.v.count[a] + 2 ? live_max : p.v.count[a] + 2;
  const char* fw = knob_env("LDM_POST_WAVE");  // A/B aid and test hook (read per launch: tests toggle it)
  const bool force_wave = fw && atoi(fw) != 0;
  const bool wave = p.logp_in || p.logp_out || live_max > 48 || force_wave;
  auto kern = wave ? (p.f32_lse ? posterior_sample_k<64, false, true> : posterior_sample_k<64, false, false>)
                   : (p.f32_lse ? posterior_sample_k<16, true, true> : posterior_sample_k<16, true, false>);
  hipLaunchKernelGGL(kern, dim3(wave ? (M + 3) / 4 : (M + 15) / 16), dim3(256), 0, st, p);
}

}  // namespace ldm
