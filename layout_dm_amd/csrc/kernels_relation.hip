// cond=relation: the logit adjustment of the reference's sampler, one workgroup per layout, the
// layout's log-probabilities in LDS for all SGD iterations (arithmetic: ldm_relation_core.h).
//
//   relation_update_k   the split-step hook (ldm_relation_update): (B,C,S) tensor in, adjusted in place
//   relation_step_k     the per-step path's fused tail of an ADJUSTED step (t >= 10): predict_start tail + q_posterior +
//                       strong mask (ldm_post_token.h) -> SGD -> [PAD] disable -> draw, logits in, tokens (and the next
//                       step's embedding rows) out — ONE launch per chunk-step where r03 had three around a 19.8 MB
//                       (chunk, S, C) log-probability tensor.  Reference order: base.py:243-291.
#include "ldm_kernels.h"
#include "ldm_relation_core.h"

namespace ldm {

static __device__ __forceinline__ RelGraph rel_graph(const RelArgs& a) {
  RelGraph gph;
  gph.edge_off = a.edge_off; gph.edge_src = a.edge_src; gph.edge_dst = a.edge_dst; gph.edge_attr = a.edge_attr;
  gph.centres = a.centres;
#pragma unroll
  for (int x = 0; x < 4; ++x) gph.canvas_bins[x] = a.canvas_bins[x];
  gph.step = a.step;
  gph.num_update = a.num_update;
  return gph;
}

__global__ __launch_bounds__(256) void relation_update_k(RelArgs a) {
  __shared__ float lgs[REL_MAX_ELEM * 4 * 32];   // logits of the bbox sub-vocabularies
  __shared__ float prs[REL_MAX_ELEM * 4 * 32];   // their softmax
  __shared__ float scratch[kRelScratchFloats];
  __shared__ int inc_off[kRelIncOffInts];
  __shared__ unsigned short inc[2 * REL_MAX_EDGE];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int E = a.S / a.A, NB = a.n_bin;
  const int e0 = a.edge_off[b], ne = a.edge_off[b + 1] - e0;
  const RelGraph gph = rel_graph(a);
  relation_incidence(gph, e0, ne, tid, E, scratch, inc_off, inc, [] { __syncthreads(); });
  relation_nodes(scratch, tid, E, a.pad_id, NB, a.centres, a.canvas_bins,
                 [&](int e) { return a.cond_seq[(size_t)b * a.S + e * a.A]; });
  __syncthreads();
  const int* node_of = reinterpret_cast<const int*>(scratch + kRelNodeOff);
  const int n_item = E * 4 * NB;
  // the lanes run along the contiguous axis of the log-probabilities: the ELEMENT index for the API's (B, C, S) layout
  // (25 reads inside one 500-byte class row instead of 64 rows per wave instruction), the bin index for (B, S, C)
  auto at = [&](int e, int x, int n) -> size_t {
    const size_t cls = (size_t)(a.n_category + x * NB + n), pos = (size_t)(e * a.A + 1 + x);
    return a.logp_tm ? ((size_t)b * a.S + pos) * a.C + cls : ((size_t)b * a.C + cls) * a.S + pos;
  };
  for (int i = tid; i < n_item; i += 256) {
    int e, x, n;
    if (a.logp_tm) { e = i / (4 * NB); x = (i / NB) % 4; n = i % NB; } else { e = i % E; x = (i / E) / NB; n = (i / E) % NB; }
    lgs[(e * 4 + x) * NB + n] = node_of[e] > 0 ? a.logp[at(e, x, n)] : 0.f;
  }
  __syncthreads();
  relation_sgd<false>(gph, e0, ne, tid, E, NB, [&](int e, int x) { return lgs + (e * 4 + x) * NB; },
                      [&](int e, int x) { return prs + (e * 4 + x) * NB; }, scratch, inc_off, inc, RelPersist{nullptr, nullptr},
                      [] { __syncthreads(); });
  for (int i = tid; i < n_item; i += 256) {
    int e, x, n;
    if (a.logp_tm) { e = i / (4 * NB); x = (i / NB) % 4; n = i % NB; } else { e = i % E; x = (i / E) / NB; n = (i / E) % NB; }
    if (node_of[e] > 0) a.logp[at(e, x, n)] = lgs[(e * 4 + x) * NB + n];
  }
}

void launch_relation_update(const RelArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.num_update <= 0) return;
  hipLaunchKernelGGL(relation_update_k, dim3(a.B), dim3(256), 0, st, a);
}

// ---- the fused tail of an adjusted step.  LDS: per token a row of kRelRowLd floats — [0, 48) the log-probabilities of its
// live classes in slot order (body bins first: the SGD's 32 logits of a bbox token are the row's first 32 floats), [48, 80)
// the softmax scratch of the SGD — then the SGD's scratch, then the samplers' scratch (2 x 48 floats per 16-lane group).
constexpr int kRelRowLd = 80;
constexpr int kRelStepMaxS = 128;
constexpr int kRelStepLds = (kRelStepMaxS * kRelRowLd + kRelScratchFloats + 16 * 96) * 4 + kRelIncBytes;

template <bool FAST>
__global__ __launch_bounds__(256) void relation_step_k(PostArgs p, RelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float rsm[];
  float* rows = rsm;
  float* scratch = rsm + kRelStepMaxS * kRelRowLd;
  float* samp = scratch + kRelScratchFloats;
  int* inc_off = reinterpret_cast<int*>(samp + 16 * 96);
  unsigned short* inc = reinterpret_cast<unsigned short*>(inc_off + kRelIncOffInts);
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int S = p.S, C = p.v.n_class, A = p.v.n_attr;
  const int E = S / A, NB = a.n_bin;
  const int grp = tid >> 4;
  const ldm_post::DppGroup<16, FAST> g{tid & 15};
  const ldm_post::SlotMap<16, 3, true> m{tid & 15};
  const RelGraph gph = rel_graph(a);
  const int e0 = a.edge_off[b], ne = a.edge_off[b + 1] - e0;
  relation_incidence(gph, e0, ne, tid, E, scratch, inc_off, inc, [] { __syncthreads(); });
  relation_nodes(scratch, tid, E, p.v.pad_id, NB, a.centres, a.canvas_bins,
                 [&](int e) { return p.cond_seq[(size_t)b * S + e * A]; });
  __syncthreads();
  const int* node_of = reinterpret_cast<const int*>(scratch + kRelNodeOff);
  // a conditioned (strong-masked) token takes its conditioned value without posterior or draw (ldm_post_token.h
  // strong_shortcut) — unless it is a bbox token of a graph node: the reference's SGD then moves that one-hot row like any
  // other (update() runs on the whole tensor AFTER the strong mask, base.py:245-269), so it goes the long way
  auto shortcut = [&](const ldm_post::TokenArgs& ta, int s) {
    return ldm_post::strong_shortcut(ta) && !(s % A != 0 && node_of[s / A] > 0);
  };

  auto token_args = [&](int s) {
    const int row = b * S + s, attr = s % A;
    ldm_post::TokenArgs ta{};
    ta.tok = p.tokens[row];
    ta.start = p.v.start[attr];
    ta.count = p.v.count[attr];
    ta.pad_id = p.v.pad_id;
    ta.mask_id = p.v.mask_id;
    ta.n_class = C;
    ta.cond_tok = p.cond_seq[row];
    ta.strong = p.strong && p.strong[row];
    ta.weak = nullptr;          // (the refinement prior belongs to another cond type)
    ta.weak_stride = S;
    ta.pad_disable = false;     // applied AFTER the adjustment (base.py:272-284), in the draw phase below
    ta.kind = p.kind;
    ta.temperature = p.temperature;
    ta.top_p = p.top_p;
    ta.top_k = p.top_k;
    ta.pos = (uint32_t)s;
    ta.step = (uint32_t)p.step;
    return ta;
  };
  // ---- phase A: log p(x_{t-1} | x_t) of every free token -> LDS
  for (int s = grp; s < S; s += 16) {
    const ldm_post::TokenArgs ta = token_args(s);
    if (shortcut(ta, s)) continue;   // conditioned token: its row is never read
    const int attr = s % A;
    const float* lrow = p.logits + (size_t)(b * S + s) * p.ldl;
    float mx = -INFINITY;
    for (int c = g.lane(); c < C - 1; c += 16) mx = fmaxf(mx, lrow[c]);
    mx = g.gmax(mx);
    float l0[3], lp[3];
    if (FAST) {
      float se = 0.f;
      for (int c = g.lane(); c < C - 1; c += 16) se += g.exp(lrow[c] - mx);
      const float lse0 = g.log(g.gsum(se));
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = m.cls(ta, j);
        l0[j] = (m.valid(ta, j) && c < C - 1) ? ldm_post::l0_f32(lrow[c], mx, lse0) : -70.0f;
      }
    } else {
      double se = 0.0;
      for (int c = g.lane(); c < C - 1; c += 16) se += exp((double)lrow[c] - (double)mx);
      const double lse0 = log(g.gsumd(se));
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = m.cls(ta, j);
        l0[j] = (m.valid(ta, j) && c < C - 1) ? ldm_post::l0_f64(lrow[c], mx, lse0) : -70.0f;
      }
    }
    const int T1 = p.T + 1, t = p.t_post, u = (t - 1 + T1) % T1;  // constrained.py:114
    auto sch = [&](int kind, int idx) { return p.sched[((size_t)kind * A + attr) * T1 + idx]; };
    const ldm_post::StepSchedule sc{sch(kLogAt, t),    sch(kLogBt, t),    sch(kLogCt, t),    sch(kLogCumAt, t),
                                    sch(kLogCumBt, t), sch(kLogCumCt, t), sch(kLogCumAt, u), sch(kLogCumBt, u),
                                    sch(kLogCumCt, u), sch(kLog1mCumCt, u)};
    ldm_post::token_log_probs(g, m, ta, sc, l0, lp);
    float* r = rows + s * kRelRowLd;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (m.valid(ta, j)) r[m.sidx(j)] = lp[j];
  }
  __syncthreads();
  // ---- phase B: the SGD on the bbox tokens of the graph's nodes (element e, coordinate x = token e A + 1 + x)
  relation_sgd<false>(gph, e0, ne, tid, E, NB, [&](int e, int x) { return rows + (e * A + 1 + x) * kRelRowLd; },
                      [&](int e, int x) { return rows + (e * A + 1 + x) * kRelRowLd + 48; }, scratch, inc_off, inc,
                      RelPersist{nullptr, nullptr}, [] { __syncthreads(); });
  // ---- phase C: [PAD] disable + draw (+ the next step's embedding row)
  for (int s = grp; s < S; s += 16) {
    ldm_post::TokenArgs ta = token_args(s);
    const int row = b * S + s, attr = s % A;
    int token;
    if (shortcut(ta, s)) {
      token = ta.cond_tok;
    } else {
      ta.pad_disable = attr != 0 && ta.cond_tok != p.v.pad_id;
      if (p.kind != 0) {
        ta.layout = p.rng[1] + (uint64_t)p.layout_off + (uint64_t)b;
        ta.seed = p.rng[0];
      }
      const float* r = rows + s * kRelRowLd;
      float lp[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) lp[j] = m.valid(ta, j) ? r[m.sidx(j)] : -INFINITY;
      ldm_post::pad_disable_only(m, ta, lp);
      float* sc_lg = samp + grp * 96;
      token = ldm_post::draw_token(g, m, ta, lp, sc_lg, sc_lg + 48, true).token;
    }
    if (g.lane() == 0) p.tokens_out[row] = token;
    if (p.x_next) {
      const float4* e = reinterpret_cast<const float4*>(p.emb + (size_t)token * p.D);
      const float4* ps = reinterpret_cast<const float4*>(p.pos + (size_t)s * p.D);
      float4* o = reinterpret_cast<float4*>(p.x_next + (size_t)row * p.ldx);
      for (int c = g.lane(); c < (p.D >> 2); c += 16) {
        const float4 x = e[c], y = ps[c];
        o[c] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
      }
    }
  }
}

// p: the step's PostArgs (logits, tokens in / out, cond_seq + strong mask, schedule, sampler, RNG, x_next); a: the graph.
// Requires S <= 128, live sub-vocabularies <= 48 classes (the caller checks and falls back to the three-launch form).
void launch_relation_step(const PostArgs& p, const RelArgs& a, hipStream_t st) {
  if (p.B <= 0) return;
  auto kern = p.f32_lse ? relation_step_k<true> : relation_step_k<false>;
  allow_big_lds((const void*)kern);
  hipLaunchKernelGGL(kern, dim3(p.B), dim3(256), kRelStepLds, st, p, a);
}
bool relation_step_supported(const PostArgs& p) {
  int live_max = 0;
  for (int at = 0; at < p.v.n_attr; ++at) live_max = live_max > p.v.count[at] + 2 ? live_max : p.v.count[at] + 2;
  return p.S <= kRelStepMaxS && live_max <= 48 && p.v.n_attr <= kMaxAttr;
}

}  // namespace ldm
