// Host side of libldm_hip.so: the hot path — one reverse step (_sample_single_step, base.py:205-291), the one-launch loop of
// the fast mode, and the per-lane hipGraph loop of the per-step path (BaseMaskAndReplaceDiffusion.sample, base.py:293-371).
#include "ldm_handle.h"

using namespace ldm_host;

void ldm_host::fill_post(ldm_handle* h, PostArgs& p, const ldm_cond* cond, const ldm_sampler* s, size_t layout_off,
                      int Bc) {
  p.sched = h->sched;
  p.f32_lse = h->cfg.precision == LDM_PREC_FAST_F16 ? 1 : 0;
  p.T = h->T;
  p.B = Bc;
  p.S = h->S;
  p.v = h->vocab;
  p.rng = h->rng;
  if (cond) {
    const size_t ro = layout_off * h->S;
    p.cond_seq = cond->d_cond_seq ? cond->d_cond_seq + ro : nullptr;
    p.strong = cond->d_strong_mask ? cond->d_strong_mask + ro : nullptr;
    p.weak = cond->d_weak_logits ? cond->d_weak_logits + layout_off * h->C * h->S : nullptr;
    p.pad_disable = cond->pad_disable;
  }
  if (s) {
    p.kind = s->kind;
    p.temperature = s->temperature;
    p.top_p = s->top_p;
    p.top_k = s->top_k;
  }
}

int ldm_host::check_ready(ldm_handle* h, int B) {
  if (!h) return -1;
  h->activate(0);
  if (!h->finalized) return h->fail(-5, "weights not finalized: call ldm_finalize_weights first");
  if (B < 1 || B > h->cfg.max_batch) return h->fail(-1, "batch %d outside [1, max_batch=%d]", B, h->cfg.max_batch);
  return 0;
}

int ldm_host::check_sampler(ldm_handle* h, const ldm_sampler* s) {
  if (!s) return h->fail(-1, "null sampler");
  if (s->kind < 0 || s->kind > 4) return h->fail(-1, "unknown sampler kind %d", s->kind);
  if (s->kind != LDM_SAMPLE_DETERMINISTIC && !(s->temperature > 0.f)) return h->fail(-1, "temperature must be > 0");
  if (s->kind == LDM_SAMPLE_TOP_P && !(s->top_p > 0.f && s->top_p <= 1.f)) return h->fail(-1, "top_p must be in (0,1]");
  if (s->kind == LDM_SAMPLE_TOP_K && (s->top_k < 1 || s->top_k > h->C)) return h->fail(-1, "top_k out of range");
  return 0;
}

static int check_relation(ldm_handle* h, const ldm_relation* rel, const ldm_cond* cond, int B) {
  if (!rel) return 0;
  if (!cond || !cond->d_cond_seq) return h->fail(-1, "cond=relation needs cond->d_cond_seq (the conditioned sequence)");
  if (!rel->d_edge_offsets || !rel->d_centres) return h->fail(-1, "ldm_relation: null edge offsets / centres");
  if (rel->n_graph_total < B) return h->fail(-1, "ldm_relation.n_graph_total smaller than B");
  if (h->cfg.max_elem > 32 || h->cfg.n_bin > 32) return h->fail(-4, "relation kernel: max_elem and n_bin must be <= 32");
  for (int x = 0; x < 4; ++x)
    if (rel->canvas_bins[x] < 0 || rel->canvas_bins[x] >= h->cfg.n_bin) return h->fail(-1, "canvas bin out of range");
  // The (chunk, C, S) log-probability buffer exists only for the three-launch form of an adjusted step (posterior ->
  // relation_update -> draw): the one-launch loop and relation_step_k keep the rows in LDS.  Allocated here, never inside
  // step_all (which may run under stream capture), and only when that form can be reached.
  PostArgs probe{};
  fill_post(h, probe, nullptr, nullptr, 0, 1);
  const bool three_launch = !loop_fusable(h, rel) && !(relation_step_supported(probe) && knob_int("LDM_REL_FUSED", 1) != 0);
  for (int l = 0; three_launch && l < h->n_lanes; ++l) {
    if (h->ws[l].rel_logp) continue;
    float* buf = nullptr;
    int rc = h->dalloc(&buf, (size_t)h->chunk * h->C * h->S, false);
    if (rc) return rc;
    h->ws[l].rel_logp = buf;
    if (l == h->cur_lane) h->rel_logp = buf;
  }
  return 0;
}

void ldm_host::fill_rel(ldm_handle* h, RelArgs& a, const ldm_relation* rel, size_t layout_off, int Bc) {
  a.edge_off = rel->d_edge_offsets + layout_off;  // offsets are absolute positions in the edge arrays
  a.edge_src = rel->d_edge_src; a.edge_dst = rel->d_edge_dst; a.edge_attr = rel->d_edge_attr;
  a.centres = rel->d_centres;
  for (int x = 0; x < 4; ++x) a.canvas_bins[x] = rel->canvas_bins[x];
  a.step = rel->relation_lambda / (14.0f * (float)rel->n_graph_total);
  a.num_update = rel->num_update; a.B = Bc; a.C = h->C; a.S = h->S; a.A = h->cfg.n_attr;
  a.n_category = h->cfg.n_category; a.n_bin = h->cfg.n_bin; a.pad_id = h->vocab.pad_id;
}

// one fused reverse step over the whole batch, chunk by chunk.  `cond` / `rel` describe layouts 0..B of THIS call
// (the loop body hands over pointers already advanced to its chunk); rel_layout_off = position of row 0 inside the
// relation graph's CSR offsets.
static int step_all(ldm_handle* h, const int32_t* tin, int32_t* tout, int t_model, int t_post, const ldm_cond* cond,
                    const ldm_relation* rel, size_t rel_layout_off, const ldm_sampler* s, int step, int B,
                    size_t rng_layout_off, hipStream_t st, bool skip_embed = false, bool embed_next = false,
                    int tie_row = -1) {
  if (t_model < 0 || t_model >= h->T || t_post < 0 || t_post >= h->T)
    return h->fail(-1, "timestep out of range [0,%d)", h->T);  // constrained.py:139
  for (int off = 0; off < B; off += h->chunk) {
    const int Bc = std::min(h->chunk, B - off);
    int rc = denoise_chunk(h, tin + (size_t)off * h->S, t_model, Bc, st, skip_embed);
    if (rc) return rc;
    PostArgs p{};
    fill_post(h, p, cond, s, off, Bc);
    p.logits = h->logits;
    p.ldl = h->Cp;
    p.tokens = tin + (size_t)off * h->S;
    p.t_post = t_post;
    p.step = step;
    p.layout_off = (int)(rng_layout_off + off);
    // cond=relation adjusts the log-probabilities only while t >= 10 (logit_adjustment.py:107); the remaining steps are
    // a plain constrained step with the [PAD] disable, i.e. the fused posterior + draw launch
    const bool adjust = rel && t_model >= 10 && rel->num_update > 0;
    if (!adjust) {
      if (rel) p.pad_disable = 1;
      p.tokens_out = tout + (size_t)off * h->S;
      if (embed_next) {  // (one chunk per call: run_loop_body)
        p.x_next = h->P; p.emb = h->emb; p.pos = h->pos; p.D = h->D; p.ldx = h->D;
      }
      if (tie_row >= 0 && h->tie_rel > 0.f && h->tie_flags && s->kind == LDM_SAMPLE_DETERMINISTIC) {
        p.tie_flags = h->tie_flags + (size_t)tie_row * h->cfg.max_batch + rng_layout_off + off;
        p.tie_rel = h->tie_rel;
        p.tie_abs = h->tie_abs;
      }
      ldm_handle::Scope sc(h, st, "posterior_sample", 0, (double)Bc * h->S * (h->Cp * 4 + 8));
      launch_posterior_sample(p, st);
      continue;
    }
    // cond=relation (base.py:243-291): posterior + strong mask -> logit adjustment -> [PAD] disable -> draw
    if (relation_step_supported(p) && knob_int("LDM_REL_FUSED", 1) != 0) {  // ... in ONE launch (r04)
      PostArgs q = p;
      q.pad_disable = 1;
      q.tokens_out = tout + (size_t)off * h->S;
      if (embed_next) {
        q.x_next = h->P; q.emb = h->emb; q.pos = h->pos; q.D = h->D; q.ldx = h->D;
      }
      RelArgs a{};
      a.cond_seq = cond->d_cond_seq + (size_t)off * h->S;
      fill_rel(h, a, rel, rel_layout_off + off, Bc);
      ldm_handle::Scope sc(h, st, "relation_step", 0, (double)Bc * h->S * (h->Cp * 4 + 8));
      launch_relation_step(q, a, st);
      continue;
    }
    if (!h->rel_logp) return h->fail(-5, "cond=relation: the three-launch step has no log-probability buffer (check_relation)");
    {
      PostArgs q = p;
      q.pad_disable = 0;  // applied after the adjustment, below
      q.logp_out = h->rel_logp;
      q.logp_tm = 1;  // (the handle's own buffer: token-major, a token's classes contiguous)
      q.tokens_out = nullptr;
      ldm_handle::Scope sc(h, st, "posterior", 0, (double)Bc * h->S * (h->Cp * 4 + h->C * 4));
      launch_posterior_sample(q, st);
    }
    {
      RelArgs a{};
      a.logp = h->rel_logp;
      a.logp_tm = 1;
      a.cond_seq = cond->d_cond_seq + (size_t)off * h->S;
      fill_rel(h, a, rel, rel_layout_off + off, Bc);
      ldm_handle::Scope sc(h, st, "relation_update", 0, (double)Bc * 4 * h->cfg.n_bin * h->cfg.max_elem * 8);
      launch_relation_update(a, st);
    }
    {
      PostArgs q{};
      fill_post(h, q, cond, s, off, Bc);
      q.strong = nullptr;  // already imposed on rel_logp
      q.weak = nullptr;
      q.pad_disable = 1;
      q.logp_in = h->rel_logp;
      q.logp_tm = 1;
      q.tokens_out = tout + (size_t)off * h->S;
      q.step = step;
      q.layout_off = (int)(rng_layout_off + off);
      if (embed_next) {  // the next step's embedding rows, as in the fused launch above
        q.x_next = h->P; q.emb = h->emb; q.pos = h->pos; q.D = h->D; q.ldx = h->D;
      }
      ldm_handle::Scope sc(h, st, "pad_disable_sample", 0, (double)Bc * h->S * (h->C * 4 + 8));
      launch_posterior_sample(q, st);
    }
  }
  return 0;
}

// ---- the whole reverse loop in one launch (kernels_stack.hip HEAD == 2) -----------------------------------------
// Eligible: fast numerics on the layout-resident kernels (the reference's backbone, S <= 128), a vocabulary of 5 head
// tiles whose attribute sub-vocabularies fit the fused tail.  cond=relation (r04): its logit adjustment couples the
// elements of a layout through an SGD on the log-probabilities — the layout's workgroup holds them in LDS behind the
// vocabulary head, so the adjusted steps run posterior -> SGD -> [PAD] disable -> draw in the same launch
// (stack_stream_k<., 2, true>); needs the constrained vocabulary with <= 32 bins and <= 32 elements.
bool ldm_host::loop_fusable(const ldm_handle* h, const ldm_relation* rel) {
  int live_max = 0;
  for (int a = 0; a < h->cfg.n_attr; ++a) live_max = std::max(live_max, h->vocab.count[a] + 2);
  if (rel && (h->rel_loop == 0 || h->cfg.q_type != LDM_Q_CONSTRAINED || h->cfg.n_bin > 32 || h->cfg.max_elem > 32 ||
              h->cfg.n_attr != 5))
    return false;
  return h->stack_loop && h->cfg.precision == LDM_PREC_FAST_F16 && h->fused_attn == 6 && h->head_img_ks && h->Cp == 160 && live_max <= kStackPostMaxLive && h->S <= 128 &&
         h->T < 32768 && !h->fast.empty() && h->tbl_att_dyn && h->D == 464 && h->F <= 2048;
}

// tokens_in -> tokens_out (may alias) through n_steps reverse steps; step0 = loop index of the first one (RNG counter
// word); cond pointers describe layout 0..B of this call; d_inter (n_steps, B, S) or nullptr; tie_row0 >= 0: near-tie
// flags of step i go to row tie_row0 + i of h->tie_flags
static int run_loop_fused(ldm_handle* h, const int32_t* tin, int32_t* tout, const ldm_cond* cond, const ldm_relation* rel,
                          const int32_t* t_model, const int32_t* t_post, int n_steps, const ldm_sampler* s, int step0, int B,
                          int32_t* d_inter, int tie_row0, hipStream_t st) {
  const int D = h->D, F = h->F, M = B * h->S;
  FusedLayerSet ls{};
  ls.n_layer = h->L;
  for (int i = 0; i < h->L; ++i) {
    const LayerW& w = h->layers[i];
    ls.w[i] = FusedLayerW{h->fast[i].attn_head_img_ks, h->fast[i].b_in, nullptr, nullptr, h->fast[i].b_out_v,
                          h->fast[i].ffn_img_pipe, w.b1, w.b2, w.g2, w.be2};
  }
  const StackHead hd{h->head_img_ks, h->head_g, h->head_b, nullptr, h->Cp, h->Cp / 32};
  const double step_flops = h->L * (gemm_flops(M, 3 * D, D) + 4.0 * B * h->H * (double)h->S * h->S * h->dh +
                                    gemm_flops(M, D, D) + 2 * gemm_flops(M, F, D)) + gemm_flops(M, h->C, D);
  for (int i0 = 0; i0 < n_steps; i0 += kStackLoopMaxSteps) {  // (timesteps travel in the kernel arguments)
    const int n = std::min(kStackLoopMaxSteps, n_steps - i0);
    PostArgs p{};
    fill_post(h, p, cond, s, 0, B);
    p.tokens = i0 == 0 ? tin : tout;
    p.tokens_out = tout;
    p.step = step0 + i0;
    p.layout_off = 0;
    p.emb = h->emb; p.pos = h->pos; p.D = D;
    RelArgs ra{};
    if (rel) {
      p.pad_disable = 1;  // cond type relation (base.py:272)
      fill_rel(h, ra, rel, 0, B);
    }
    if (tie_row0 >= 0 && h->tie_rel > 0.f && h->tie_flags && s->kind == LDM_SAMPLE_DETERMINISTIC) {
      p.tie_flags = h->tie_flags + (size_t)(tie_row0 + i0) * h->cfg.max_batch;
      p.tie_rel = h->tie_rel;
      p.tie_abs = h->tie_abs;
    }
    StackLoop lp{};
    lp.tables = StackTables{h->tbl_att_static, h->tbl_att_dyn, h->tbl_ffn, h->tbl_head};
    lp.post = &p; lp.adaln = h->adaln; lp.t_model = t_model + i0; lp.t_post = t_post + i0;
    lp.inter = d_inter ? d_inter + (size_t)i0 * B * h->S : nullptr;
    lp.n_steps = n; lp.inter_ld = B; lp.tie_ld = h->cfg.max_batch;
    lp.rel = rel ? &ra : nullptr;
    ldm_handle::Scope sc(h, st, "layers_fused_loop", n * step_flops, (double)B * h->S * 8);
    launch_stack_loop(ls, F, D, B, h->S, h->H, h->dh, hd, lp, st);
  }
  return 0;
}

// clears the rows a deterministic call is about to fill
static int tie_begin(ldm_handle* h, const ldm_sampler* s, const ldm_relation* rel, int n_steps, int B, hipStream_t st) {
  (void)B;
  if (!(h->tie_rel > 0.f) || !h->tie_flags || s->kind != LDM_SAMPLE_DETERMINISTIC) return 0;
  // the adjusted steps of cond=relation draw from the SGD's output, where the lead of the winner is no longer a
  // function of the logits with a known Lipschitz bound: no report exists for them, so none may be assumed
  if (rel) return h->fail(-1, "near-tie report is not defined for cond=relation: decode in LDM_PREC_EXACT_F32 instead");
  if (n_steps > h->tie_steps) return h->fail(-1, "near-tie report: at most %d steps per call", h->tie_steps);
  HIP_OK(h, hipMemsetAsync(h->tie_flags, 0, (size_t)n_steps * h->cfg.max_batch, st));
  return 0;
}

// ------------------------------------------------------------------------------------------ hot path
extern "C" int ldm_sample_step(ldm_handle* h, const int32_t* d_tokens_in, int32_t* d_tokens_out, int t_model,
                               int t_post, const ldm_cond* cond, const ldm_relation* rel, const ldm_sampler* s,
                               uint64_t seed, uint64_t first_layout, int step, int B, void* stream) {
  int rc = check_ready(h, B);
  if (rc) return rc;
  if ((rc = check_sampler(h, s))) return rc;
  if (!d_tokens_in || !d_tokens_out) return h->fail(-1, "null argument");
  ON_DEVICE(h);
  if ((rc = check_relation(h, rel, cond, B))) return rc;
  hipStream_t st = (hipStream_t)stream;
  if ((rc = set_rng(h, seed, first_layout, st))) return rc;
  if (t_model < 0 || t_model >= h->T || t_post < 0 || t_post >= h->T)
    return h->fail(-1, "timestep out of range [0,%d)", h->T);  // constrained.py:139
  if ((rc = tie_begin(h, s, rel, 1, B, st))) return rc;
  if (loop_fusable(h, rel)) {
    const int32_t tm = t_model, tp = t_post;
    if ((rc = run_loop_fused(h, d_tokens_in, d_tokens_out, cond, rel, &tm, &tp, 1, s, step, B, nullptr, 0, st))) return rc;
  } else if ((rc = step_all(h, d_tokens_in, d_tokens_out, t_model, t_post, cond, rel, 0, s, step, B, 0, st, false, false, 0))) {
    return rc;
  }
  HIP_OK(h, hipGetLastError());
  return 0;
}

// The T-step loop of the chunks of ONE lane (lane < 0: every chunk, in order, through lane 0's workspace).
static int run_loop_body(ldm_handle* h, const ldm_cond* cond, const ldm_relation* rel, const int32_t* t_model,
                         const int32_t* t_post, int n_steps, const ldm_sampler* s, int B, int32_t* d_inter,
                         int lane, hipStream_t st) {
  // state lives in tok_a / tok_b (ping-pong); chunk-major order keeps one chunk's activations and the
  // weights resident in L2 / Infinity Cache for all T steps before moving to the next chunk
  const size_t S = h->S;
  // BALANCED chunks (r06): a call of B layouts needs ceil(B / chunk) passes; they share the layouts evenly — 300 layouts are 150 + 150, not 256 + 44
  // (the 44-layout pass kept 212 of 256 compute units idle for as long as the full one; a call's two passes also overlap on the lanes when each
  // fills only part of the chip: same-box split 1 012 -> 1 105, hybrid 1 818 -> 2 011 layouts/s at B = 300; + 6 % / + 1 % at B = 400).  Tokens do not depend on the cut (tests/test_config34_shapes.py).
  // Only where the remainder is small: with a remainder of >= 3/4 of a chunk the uneven cut is 2 % FASTER for the short hybrid launches (488 = 256 + 232: 2 142
  // against 2 093 for 244 + 244 — passes of unequal length drift against each other on the lanes instead of running in lock-step; profiles/r06_call16_17_25_*).
  const int n_pass = (B + h->chunk - 1) / h->chunk;
  const int rem = B - (n_pass - 1) * h->chunk;
  const int cb = (h->balanced_chunks && n_pass > 1 && 4 * rem < 3 * h->chunk) ? (B + n_pass - 1) / n_pass : h->chunk;
  const int first = lane < 0 ? 0 : lane * cb;
  const int stride = lane < 0 ? cb : h->n_lanes * cb;
  h->activate(lane < 0 ? 0 : lane);
  for (int off = first; off < B; off += stride) {
    const int Bc = std::min(cb, B - off);
    ldm_cond cc{};
    if (cond) {
      cc = *cond;
      if (cc.d_cond_seq) cc.d_cond_seq += off * S;
      if (cc.d_strong_mask) cc.d_strong_mask += off * S;
      if (cc.d_weak_logits) cc.d_weak_logits += (size_t)off * h->C * S;
    }
    int32_t* cur = h->tok_a + off * S;
    int32_t* nxt = h->tok_b + off * S;
    // the stack kernel takes raw rows and computes its own row statistics, so the posterior kernel of step i can write
    // step i + 1's embedding itself (no separate embedding launch inside the loop)
    const bool fuse_embed = h->cfg.precision == LDM_PREC_FAST_F16 && h->fused_attn == 6;
    for (int i = 0; i < n_steps; ++i) {
      int rc = step_all(h, cur, nxt, t_model[i], t_post[i], cond ? &cc : nullptr, rel, off, s, i, Bc, off, st,
                        fuse_embed && i > 0, fuse_embed && i + 1 < n_steps, i);
      if (rc) return rc;
      if (d_inter)
        HIP_OK(h, hipMemcpyAsync(d_inter + ((size_t)i * B + off) * S, nxt, (size_t)Bc * S * 4,
                                 hipMemcpyDeviceToDevice, st));
      std::swap(cur, nxt);
    }
    if (n_steps % 2 == 1)  // result sits in tok_b: bring it back to tok_a
      HIP_OK(h, hipMemcpyAsync(h->tok_a + off * S, h->tok_b + off * S, (size_t)Bc * S * 4, hipMemcpyDeviceToDevice, st));
  }
  return 0;
}

extern "C" int ldm_sample_loop(ldm_handle* h, int32_t* d_tokens_inout, const ldm_cond* cond, const ldm_relation* rel,
                               const int32_t* h_t_model, const int32_t* h_t_post, int n_steps, const ldm_sampler* s,
                               uint64_t seed, uint64_t first_layout, int B, int32_t* d_intermediates, int use_graph,
                               void* stream) {
  int rc = check_ready(h, B);
  if (rc) return rc;
  if ((rc = check_sampler(h, s))) return rc;
  if (!d_tokens_inout || !h_t_model || !h_t_post || n_steps < 1) return h->fail(-1, "bad argument");
  for (int i = 0; i < n_steps; ++i)
    if (h_t_model[i] < 0 || h_t_model[i] >= h->T || h_t_post[i] < 0 || h_t_post[i] >= h->T)
      return h->fail(-1, "timestep out of range [0,%d)", h->T);
  ON_DEVICE(h);
  if ((rc = check_relation(h, rel, cond, B))) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t nbytes = (size_t)B * h->S * 4;
  HIP_OK(h, hipEventRecord(h->loop_a, st));
  if ((rc = set_rng(h, seed, first_layout, st))) return rc;
  if ((rc = tie_begin(h, s, rel, n_steps, B, st))) return rc;
  if (loop_fusable(h, rel)) {
    // one launch: every layout's workgroup runs all its steps in place on the caller's tokens (no staging, no graph)
    if ((rc = run_loop_fused(h, d_tokens_inout, d_tokens_inout, cond, rel, h_t_model, h_t_post, n_steps, s, 0, B,
                             d_intermediates, 0, st)))
      return rc;
    HIP_OK(h, hipEventRecord(h->loop_b, st));
    h->loop_timed = true;
    HIP_OK(h, hipGetLastError());
    return 0;
  }
  HIP_OK(h, hipMemcpyAsync(h->tok_a, d_tokens_inout, nbytes, hipMemcpyDeviceToDevice, st));
  if (use_graph && !h->profiling) {
    // copy the constraints into handle-owned staging buffers: the captured graph then only ever sees
    // fixed addresses and is reused across batches whose cond tensors live elsewhere
    ldm_cond staged{};
    if (cond) {
      const size_t nS = (size_t)B * h->S;
      staged.pad_disable = cond->pad_disable;
      if (cond->d_cond_seq) {
        HIP_OK(h, hipMemcpyAsync(h->st_cond_seq, cond->d_cond_seq, nS * 4, hipMemcpyDeviceToDevice, st));
        staged.d_cond_seq = h->st_cond_seq;
      }
      if (cond->d_strong_mask) {
        HIP_OK(h, hipMemcpyAsync(h->st_strong, cond->d_strong_mask, nS, hipMemcpyDeviceToDevice, st));
        staged.d_strong_mask = h->st_strong;
      }
      if (cond->d_weak_logits) {
        if (!h->st_weak && (rc = h->dalloc(&h->st_weak, (size_t)h->cfg.max_batch * h->C * h->S, false))) return rc;
        HIP_OK(h, hipMemcpyAsync(h->st_weak, cond->d_weak_logits, nS * h->C * 4, hipMemcpyDeviceToDevice, st));
        staged.d_weak_logits = h->st_weak;
      }
      cond = &staged;
    }
    // same for the relation graph: CSR offsets are read back once (host) to size the edge staging
    ldm_relation staged_rel{};
    if (rel) {
      staged_rel = *rel;
      std::vector<int32_t> off(B + 1);
      HIP_OK(h, hipMemcpyAsync(off.data(), rel->d_edge_offsets, (size_t)(B + 1) * 4, hipMemcpyDeviceToHost, st));
      HIP_OK(h, hipStreamSynchronize(st));
      const int32_t e0 = off[0], ne = off[B] - off[0];
      if (ne < 0) return h->fail(-1, "ldm_relation: edge offsets are not monotonic");
      if (!h->st_rel_off && (rc = h->dalloc(&h->st_rel_off, (size_t)h->cfg.max_batch + 1))) return rc;
      if (!h->st_rel_centres && (rc = h->dalloc(&h->st_rel_centres, (size_t)4 * h->cfg.n_bin))) return rc;
      if ((size_t)ne > h->st_rel_cap) {
        // a grown buffer has a new address: graphs keyed on the old one can never hit again — drop them and release
        // the old staging buffer, once NOTHING on the device can still be reading it (an earlier replay may run on
        // another stream than the one synchronised above)
        int32_t* old = h->st_rel_edges;
        if (old) HIP_OK(h, hipDeviceSynchronize());
        const size_t cap = std::max<size_t>(1024, (size_t)ne * 2);
        if ((rc = h->dalloc(&h->st_rel_edges, 3 * cap))) return rc;
        h->st_rel_cap = cap;
        if (old) {
          for (size_t gi = h->graphs.size(); gi-- > 0;)
            if (h->graphs[gi].key.rel_edges == old) {
              h->graphs[gi].destroy();
              h->graphs.erase(h->graphs.begin() + gi);
            }
          h->owned.erase(std::remove(h->owned.begin(), h->owned.end(), (void*)old), h->owned.end());
          (void)hipFree(old);
        }
      }
      for (auto& o : off) o -= e0;
      HIP_OK(h, hipMemcpyAsync(h->st_rel_off, off.data(), (size_t)(B + 1) * 4, hipMemcpyHostToDevice, st));
      HIP_OK(h, hipStreamSynchronize(st));  // `off` is pageable host memory
      if (ne > 0) {
        const size_t cap = h->st_rel_cap;
        HIP_OK(h, hipMemcpyAsync(h->st_rel_edges, rel->d_edge_src + e0, (size_t)ne * 4, hipMemcpyDeviceToDevice, st));
        HIP_OK(h, hipMemcpyAsync(h->st_rel_edges + cap, rel->d_edge_dst + e0, (size_t)ne * 4, hipMemcpyDeviceToDevice, st));
        HIP_OK(h, hipMemcpyAsync(h->st_rel_edges + 2 * cap, rel->d_edge_attr + e0, (size_t)ne * 4, hipMemcpyDeviceToDevice, st));
      }
      HIP_OK(h, hipMemcpyAsync(h->st_rel_centres, rel->d_centres, (size_t)4 * h->cfg.n_bin * 4, hipMemcpyDeviceToDevice, st));
      staged_rel.d_edge_offsets = h->st_rel_off;
      staged_rel.d_edge_src = h->st_rel_edges;
      staged_rel.d_edge_dst = h->st_rel_edges + h->st_rel_cap;
      staged_rel.d_edge_attr = h->st_rel_edges + 2 * h->st_rel_cap;
      staged_rel.d_centres = h->st_rel_centres;
      rel = &staged_rel;
    }
    GraphKey key{};
    key.B = B; key.n_steps = n_steps; key.kind = s->kind; key.top_k = s->top_k;
    key.temperature = s->temperature; key.top_p = s->top_p;
    key.has_cond = cond != nullptr;
    key.cond_seq = cond ? cond->d_cond_seq : nullptr;
    key.strong = cond ? cond->d_strong_mask : nullptr;
    key.weak = cond ? cond->d_weak_logits : nullptr;
    key.pad_disable = cond ? cond->pad_disable : 0;
    // intermediates are captured into a handle-owned buffer (fixed address) and copied out after the launch, so
    // get_intermediate_results=True replays the same graph instead of re-capturing for every caller pointer
    int32_t* inter_dst = nullptr;
    if (d_intermediates) {
      if (n_steps > h->T) return h->fail(-1, "intermediates: n_steps %d > T %d", n_steps, h->T);
      if (!h->st_inter && (rc = h->dalloc(&h->st_inter, (size_t)h->T * h->cfg.max_batch * h->S, false))) return rc;
      inter_dst = h->st_inter;
    }
    key.has_inter = inter_dst != nullptr;
    key.tie_rel = (h->tie_flags && s->kind == LDM_SAMPLE_DETERMINISTIC) ? h->tie_rel : 0.f;
    key.tie_abs = (h->tie_flags && s->kind == LDM_SAMPLE_DETERMINISTIC) ? h->tie_abs : 0.f;
    if (rel) {
      key.has_rel = 1;
      key.rel_num_update = rel->num_update;
      key.rel_n_graph = rel->n_graph_total;
      key.rel_lambda = rel->relation_lambda;
      key.rel_edges = rel->d_edge_src;
      for (int x = 0; x < 4; ++x) key.rel_bins[x] = rel->canvas_bins[x];
    }
    key.t_model.assign(h_t_model, h_t_model + n_steps);
    key.t_post.assign(h_t_post, h_t_post + n_steps);
    GraphEntry* ge = nullptr;
    for (auto& g : h->graphs)
      if (g.key == key) ge = &g;
    // lanes that actually own a chunk of this call
    const int n_chunks = (B + h->chunk - 1) / h->chunk;
    const int lanes = std::min(h->n_lanes, n_chunks);
    if (!ge) {
      if (h->graphs.size() >= 8) {  // small LRU-less cache: drop the oldest
        h->graphs[0].destroy();
        h->graphs.erase(h->graphs.begin());
      }
      GraphEntry ne;
      ne.key = key;
      for (int lane = 0; lane < lanes; ++lane) {
        // capture on a private stream so the caller's stream state is untouched; one linear graph per lane
        hipStream_t cap = nullptr;
        HIP_OK(h, hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        HIP_OK(h, hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
        if (lane > 0 && h->lane_offset_us > 0) launch_delay_us(lane * h->lane_offset_us, cap);
        rc = run_loop_body(h, cond, rel, h_t_model, h_t_post, n_steps, s, B, inter_dst, lanes > 1 ? lane : -1, cap);
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(cap, &graph);
        (void)hipStreamDestroy(cap);
        if (rc || e != hipSuccess) {
          if (graph) (void)hipGraphDestroy(graph);
          ne.destroy();
          if (rc) return rc;
          return h->fail(-2, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        ne.graph.push_back(graph);
        if (e != hipSuccess) {
          ne.destroy();
          return h->fail(-2, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
        }
        ne.exec.push_back(exec);
      }
      h->graphs.push_back(ne);
      ge = &h->graphs.back();
    }
    // lane 0 replays on the caller's stream, the others on their own streams between a fork and a join event
    if (ge->exec.size() > 1) HIP_OK(h, hipEventRecord(h->fork_ev, st));
    for (size_t lane = 1; lane < ge->exec.size(); ++lane) {
      HIP_OK(h, hipStreamWaitEvent(h->lane_stream[lane], h->fork_ev, 0));
      HIP_OK(h, hipGraphLaunch(ge->exec[lane], h->lane_stream[lane]));
      HIP_OK(h, hipEventRecord(h->lane_done[lane], h->lane_stream[lane]));
    }
    HIP_OK(h, hipGraphLaunch(ge->exec[0], st));
    for (size_t lane = 1; lane < ge->exec.size(); ++lane) HIP_OK(h, hipStreamWaitEvent(st, h->lane_done[lane], 0));
    if (inter_dst)
      HIP_OK(h, hipMemcpyAsync(d_intermediates, inter_dst, (size_t)n_steps * B * h->S * 4, hipMemcpyDeviceToDevice, st));
  } else {
    if ((rc = run_loop_body(h, cond, rel, h_t_model, h_t_post, n_steps, s, B, d_intermediates, -1, st))) return rc;
  }
  HIP_OK(h, hipMemcpyAsync(d_tokens_inout, h->tok_a, nbytes, hipMemcpyDeviceToDevice, st));
  HIP_OK(h, hipEventRecord(h->loop_b, st));
  h->loop_timed = true;
  HIP_OK(h, hipGetLastError());
  return 0;
}

