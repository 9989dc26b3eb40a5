// Host side of libldm_hip.so: the C-ABI declared in include/ldm_hip.h — lifecycle, parity hooks, result packaging, near-tie
// report, introspection.  (Weights: ldm_weights.cpp; one denoiser pass: ldm_denoise.cpp; the hot path: ldm_loop.cpp.)
// No torch types cross this boundary.
#include "ldm_handle.h"

using namespace ldm_host;

std::string& ldm_host::create_error() {
  static thread_local std::string e;
  return e;
}

// ------------------------------------------------------------------------------------------ create
extern "C" int ldm_abi_version(void) { return LDM_ABI_VERSION; }

extern "C" int ldm_get_layout(const ldm_handle* h, int* chunk, int* lanes) {
  if (!h) return -1;
  if (chunk) *chunk = h->chunk;
  if (lanes) *lanes = h->n_lanes;
  return 0;
}

extern "C" const char* ldm_last_error(const ldm_handle* h) { return h ? h->err.c_str() : create_error().c_str(); }

extern "C" int ldm_create(const ldm_config* cfg_in, int device, ldm_handle** out) {
  auto bad = [&](const char* msg) {
    create_error() = msg;
    return -1;
  };
  if (!cfg_in || !out) return bad("null argument");
  // LDM_PREC_MIXED_F16 / LDM_PREC_HYBRID_F16 ARE the split mode — same launches, images, workspace — with the weight GEMMs in their two-product
  // form (hybrid: the FFN and the head in plain fp16): from here on the handle is a split-mode handle with `mixed` set
  ldm_config cfg_local = *cfg_in;
  const int mixed = cfg_local.precision == LDM_PREC_MIXED_F16 ? 1 : cfg_local.precision == LDM_PREC_HYBRID_F16 ? 2 : 0;
  if (mixed) cfg_local.precision = LDM_PREC_SPLIT_F16;
  const ldm_config* cfg = &cfg_local;
  if (cfg->abi_version != LDM_ABI_VERSION) return bad("ldm_config.abi_version mismatch");
  if (cfg->n_attr != 5) return bad("only the c-x-y-w-h (5 attribute) vocabulary is supported");
  if (cfg->n_category < 1 || cfg->n_bin < 1 || cfg->n_layer < 1 || cfg->n_step < 1 || cfg->n_head < 1 || cfg->d_model < 16)
    return bad("n_category / n_bin / n_layer / n_step / n_head must be >= 1, d_model >= 16");
  if (cfg->n_category + 4 * cfg->n_bin + 2 > 192)
    return bad("vocabulary > 192 classes not supported by the fused posterior kernel");
  if (cfg->d_model % 16 || cfg->d_ff % 16 || cfg->d_model > 1024) return bad("d_model/d_ff must be multiples of 16, d_model <= 1024");
  if (cfg->d_model % cfg->n_head) return bad("d_model must be divisible by n_head");
  if (cfg->d_model / cfg->n_head > 64) return bad("head_dim > 64 not supported");
  if (cfg_in->precision < 0 || cfg_in->precision > 4) return bad("unknown precision mode");
  if (cfg->max_batch < 1) return bad("max_batch must be >= 1");
  {  // development knobs are refused outside dev mode (ldm_knobs.h): no stray variable changes the shipped path
    static std::string msg;
    const std::string k = knob_refused();
    if (!k.empty()) {
      msg = "environment knob " + k + " is set but LDM_DEV=1 is not: development / ablation paths are refused";
      return bad(msg.c_str());
    }
  }
  {
    // sequence length: the fp16 attention kernels hold a layout's scores in ONE 128 x 128 tile; the fp32 row kernel
    // keeps K and V of a (layout, head) in LDS (2 x S x head_dim floats <= 160 KiB).  The reference's datasets: S = 125.
    const int S = cfg->max_elem * cfg->n_attr, dh = cfg->d_model / cfg->n_head;
    if (S < 1) return bad("max_elem must be >= 1");
    if (cfg->precision == LDM_PREC_FAST_F16 && S > 128)
      return bad("precision fast: at most 128 tokens per layout (max_elem * n_attr); use precision exact");
    if ((size_t)2 * S * ((dh + 3) & ~3) * sizeof(float) > 160 * 1024)
      return bad("sequence too long for the attention kernels (2 * S * head_dim floats must fit 160 KiB of LDS)");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return bad("no HIP device visible: the MI355X path has no CPU fallback");
  if (device < 0 || device >= ndev) return bad("device index out of range");
  DeviceGuard create_guard(device);
  if (create_guard.err != hipSuccess) return bad("hipSetDevice failed");

  auto* h = new ldm_handle();
  h->cfg = *cfg;
  h->device = device;
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) h->n_cu = ncu;
  }
  h->S = cfg->max_elem * cfg->n_attr;
  h->C = cfg->n_category + 4 * cfg->n_bin + 2;
  h->D = cfg->d_model;
  h->F = cfg->d_ff;
  h->H = cfg->n_head;
  h->dh = h->D / h->H;
  h->L = cfg->n_layer;
  h->T = cfg->n_step;
  // K axes of the fp16 operand copies: whole 128-byte rows per 64-wide K tile in the split mode (the LDS-DMA split GEMM reads
  // operand rows one K tile at a time: a 64-byte row segment is half a cache line and doubles the L2 traffic)
  h->Dp = round_up(h->D, cfg->precision == LDM_PREC_SPLIT_F16 ? 64 : 32);
  h->Fp = round_up(h->F, cfg->precision == LDM_PREC_SPLIT_F16 ? 64 : 32);
  h->Cp = round_up(h->C, 32);
  // vocabulary geometry: helpers/layout_tokenizer.py:79-82,429-467
  h->vocab.n_class = h->C;
  h->vocab.n_attr = cfg->n_attr;
  h->vocab.pad_id = h->C - 2;
  h->vocab.mask_id = h->C - 1;
  if (cfg->q_type != LDM_Q_CONSTRAINED && cfg->q_type != LDM_Q_VANILLA) {
    delete h;
    return bad("unknown q_type");
  }
  for (int a = 0; a < cfg->n_attr; ++a) {
    if (cfg->q_type == LDM_Q_VANILLA) {  // one vocabulary: every class is live at every position (vanilla.py)
      h->vocab.start[a] = 0;
      h->vocab.count[a] = h->C - 2;
    } else {
      h->vocab.start[a] = (a == 0) ? 0 : cfg->n_category + (a - 1) * cfg->n_bin;
      h->vocab.count[a] = (a == 0) ? cfg->n_category : cfg->n_bin;
    }
  }
  // chunk: layouts per pass. auto = keep (x, qkv, hidden ...) of one chunk well inside the 256 MiB MALL
  int chunk = cfg->chunk;
  if (const char* ce = knob_env("LDM_CHUNK")) chunk = atoi(ce);  // experiments
  // auto: 256 layouts (M = 32 000 rows = 250 row blocks / 256 per-layout workgroups: one full round of the 256 CUs;
  // P + Q of a chunk = 119 MB stay inside the 256 MiB Infinity Cache) and two lanes (see Workspace)
  if (chunk <= 0) chunk = 256;
  chunk = std::min(chunk, cfg->max_batch);
  h->chunk = chunk;
  h->balanced_chunks = knob_int("LDM_BALANCED_CHUNKS", 1) != 0;

  // lanes: LDM_LANES / LDM_LANE_OFFSET_US override cfg->lanes (experiments); more lanes than chunks make no sense
  h->n_lanes = cfg->lanes > 0 ? cfg->lanes : 2;
  if (const char* le = knob_env("LDM_LANES")) h->n_lanes = std::max(1, atoi(le));
  h->n_lanes = std::min(h->n_lanes, std::max(1, (cfg->max_batch + chunk - 1) / chunk));
  h->lane_offset_us = 50;
  if (const char* lo = knob_env("LDM_LANE_OFFSET_US")) h->lane_offset_us = std::max(0, atoi(lo));
  h->ws.resize(h->n_lanes);
  const size_t Mc = (size_t)chunk * h->S;
  int rc = 0;
  auto A = [&](auto** p, size_t n) {
    if (rc == 0) rc = h->dalloc(p, n);
  };
  for (int lane = 0; lane < h->n_lanes; ++lane) {
  // P / Q carry padding rows up to the next multiple of 256: the row-stationary kernels write whole 128-row
  // blocks (rows >= M land in the padding instead of being exec-masked)
  const size_t Mrows = (size_t)round_up((int)Mc, 256);
  A(&h->P, Mrows * h->D);
  A(&h->Q, Mrows * h->D);
  A(&h->logits, Mc * h->Cp);
  if (cfg->precision == LDM_PREC_EXACT_F32) {
    A(&h->qkv32, Mc * 3 * h->D);
    A(&h->att32, Mc * h->D);
    A(&h->h32, Mc * h->D);
    A(&h->hid32, Mc * h->F);
  } else if (cfg->precision == LDM_PREC_FAST_F16) {
    h->Dq = round_up(h->D, 64);
    h->HD = h->H * 64;
    h->Fq = round_up(h->F, 64);
    h->Mpad = round_up((int)Mc, 256);
    const size_t Mp = h->Mpad;
    const char* env = knob_env("LDM_GEMM_CFG");  // "q,o,1,2,h" tile-config ids (tuning override)
    int defaults[5] = {5, 5, 5, 5, 5};
    for (int i = 0; i < 5; ++i) h->gemm_cfg[i] = defaults[i];
    if (env) sscanf(env, "%d,%d,%d,%d,%d", &h->gemm_cfg[0], &h->gemm_cfg[1], &h->gemm_cfg[2], &h->gemm_cfg[3], &h->gemm_cfg[4]);
    if (const char* fa = knob_env("LDM_FUSED_ATTN")) h->fused_attn = atoi(fa) == 0 ? 0 : 6;
    if (const char* sp = knob_env("LDM_STACK_LOOP")) h->stack_loop = atoi(sp);
    h->rel_loop = knob_int("LDM_REL_LOOP", 1);
    // stack kernel: one 128-row tile per layout, and every one of its 4 waves must own at least one real row (its
    // exec-masked stores are counted by the vmcnt waits) => 96 < S <= 128; d_model 464 in 8 heads, K padded to 512
    if (h->S > 128 || h->S <= 96 || h->dh > 64 || h->D != 464 || h->Dq != 512 || h->HD != 512 || h->H != 8 ||
        h->F % 32 || h->F > 2048 || h->L > 8 || h->Cp % 32)
      h->fused_attn = 0;
    A(&h->att16, Mp * h->HD);
    A(&h->qkv16, Mp * 3 * h->HD);
    A(&h->stats_a, Mp);
    A(&h->stats_b, Mp);
    if (h->fused_attn == 0) {  // LayerNorm outputs and the FFN hidden activation only exist on the generic path
      A(&h->a16, Mp * h->Dq);
      A(&h->h16, Mp * h->Dq);
      A(&h->hid16, Mp * h->Fq);
    }
  } else {
    // (whole 128-row tiles: the LDS-DMA split GEMM loads its operands without bounds checks)
    const size_t Mt = (size_t)round_up((int)Mc, 256);
    A(&h->a16, Mt * h->Dp);
    A(&h->att16, Mt * h->Dp);
    A(&h->h16, Mt * h->Dp);
    A(&h->hid16, (Mt + 64) * h->Fp);     // (+ 64 rows: the panel-major form of the hidden activations, rows rounded up to 64)
    {
      A(&h->qkv32, Mc * 3 * h->D);
      A(&h->a16lo, Mt * h->Dp);
      A(&h->att16lo, Mt * h->Dp);
      A(&h->h16lo, Mt * h->Dp);
      A(&h->hid16lo, (Mt + 64) * h->Fp);
    }
    // panel rows: the fused attention kernel reads 128 keys / query rows from a layout's first row whatever S is — the last layout of
    // a chunk reaches 128 - S rows past the chunk's last row (r06 fix: the slack was the reference's 3 rows + 5; at S = 50 the last
    // panel was over-read by 8 rows, a memory fault whenever the allocation ended on a page boundary)
    if (cfg->precision == LDM_PREC_SPLIT_F16) h->panel_rows = (size_t)round_up((int)Mc + std::max(8, 128 - h->S), 64);
    if (cfg->precision == LDM_PREC_SPLIT_F16 && attnout16x3_supported(h->S, h->H, h->dh, h->D)) {
      // q / k / v panels of the fused attention + out_proj kernel: 3 x 8 heads x 2 panels of 32 halfs, hi and lo; a layout's last
      // key tile reads up to 3 rows past the chunk's last row (slack, zero)
      A(&h->qkvp_hi, (size_t)48 * h->panel_rows * 32);
      A(&h->qkvp_lo, (size_t)48 * h->panel_rows * 32);
    }
  }
  h->save_ws(lane);
  }
  if (cfg->precision == LDM_PREC_SPLIT_F16) {
    // the LayerNorm-fed GEMMs as row-resident launches (kernels_lngemm.hip): d_model 464 (29 k16-steps, K padded to 512),
    // every N within 2048 columns; LDM_DEV=1 LDM_X3_LNGEMM=0: LayerNorm launches + gemm16x3_k (the r04 structure)
    auto even = [](int t) { return (t + 1) & ~1; };
    h->x3_qkv_tiles = even((3 * h->D + 31) / 32);
    h->x3_ffn1_tiles = even((h->F + 31) / 32);
    h->x3_head_tiles = even(h->Cp / 32);
    h->lngemm = h->D == 464 && h->Dp == 512 && (3 * h->D) % 4 == 0 && h->F % 4 == 0 && h->x3_qkv_tiles * 32 <= 2048 &&
                h->x3_ffn1_tiles * 32 <= 2048 && h->x3_head_tiles * 32 <= 2048 && h->x3_qkv_tiles * 32 <= round_up(3 * h->D, 256) &&
                h->x3_ffn1_tiles * 32 <= round_up(h->F, 256) && h->x3_head_tiles * 32 <= round_up(h->C, 256) &&
                knob_int("LDM_X3_LNGEMM", 4) != 0;
    // The two N = d_model GEMMs can run as the GEMM PROLOGUE of the row-resident kernel that normalises their sum (kernels_lngemm.hip PRE)
    // instead of as gemm16x3_k launches.  Default (level 4): linear2 only — the fused launch is 28 us per layer cheaper than the two it
    // replaces, +4 % whole-job same-box.  Fusing out_proj as well (LDM_DEV=1 LDM_X3_LNGEMM=2; 3 = out_proj only) saves nothing per launch
    // and costs the overlap between the two chunk pipelines (a fused launch fills every CU alone): -5 % (profiles/r05_call24_31_*).
    // LDM_X3_LNGEMM=1: no prologue (the first r05 structure).  The K-slab weight images are only built for what is selected.
    const int lv = (int)knob_int("LDM_X3_LNGEMM", 4);
    h->lngemm_pre = h->lngemm && h->Dp % 32 == 0 && h->Fp % 32 == 0 && h->D <= 480 && lv >= 2;
    h->pre_out = h->lngemm_pre && (lv == 2 || lv == 3);
    h->pre_ffn2 = h->lngemm_pre && (lv == 2 || lv == 4);
    // r06: attention + out_proj as one layout-resident launch behind an in_proj that writes hi / lo panels (kernels_attnout.hip);
    // LDM_DEV=1 LDM_X3_ATTNOUT=0: attn16x3_k + the out_proj launch of gemm16x3_k (the r05 structure)
    h->hid_panels = h->lngemm && h->pre_ffn2 && h->Fp % 32 == 0 && h->panel_rows <= (size_t)round_up((int)Mc, 256) + 64 &&
                    knob_int("LDM_X3_HIDPANEL", 1) != 0;
    h->attnout = h->lngemm && !h->pre_out && h->qkvp_hi && h->panel_rows * 64 * 2 < (1ull << 32) && knob_int("LDM_X3_ATTNOUT", 1) != 0;
    h->mixed = mixed;
    h->w2p = h->lngemm && h->pre_ffn2 && h->attnout && mixed && (mixed == 1 || h->hid_panels);
    if (h->w2p) {
      h->np_w = 2;
      h->np_ffn = mixed == 2 ? 1 : 2;
      h->ffn_fused = mixed == 2 && h->F % 32 == 0 && h->F <= 2048 && knob_int("LDM_HYB_FFN", 1) != 0;
      h->attn_ffn_fused = h->ffn_fused && knob_int("LDM_HYB_ATTNFFN", 1) != 0;
    }
    if (mixed && !h->w2p) {
      h->err = "precision mixed / hybrid: only the reference backbone's geometry (d_model 464, 8 heads, <= 128 tokens per layout) has the two-product kernels; use precision split";
      rc = -1;
    }
  }
  h->cur_lane = h->n_lanes - 1;
  h->activate(0);
  A(&h->tok_a, (size_t)cfg->max_batch * h->S);
  A(&h->tok_b, (size_t)cfg->max_batch * h->S);
  A(&h->st_cond_seq, (size_t)cfg->max_batch * h->S);
  A(&h->st_strong, (size_t)cfg->max_batch * h->S);
  A(&h->rng, 2);
  A(&h->sched, (size_t)kNumSched * cfg->n_attr * (h->T + 1));
  if (rc != 0) {
    create_error() = h->err;
    ldm_destroy(h);
    return rc;
  }
  if (hipEventCreate(&h->loop_a) != hipSuccess || hipEventCreate(&h->loop_b) != hipSuccess ||
      hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming) != hipSuccess) {
    create_error() = "event creation failed";
    ldm_destroy(h);
    return -2;
  }
  h->lane_stream.assign(h->n_lanes, nullptr);
  h->lane_done.assign(h->n_lanes, nullptr);
  for (int l = 1; l < h->n_lanes; ++l) {
    if (hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->lane_done[l], hipEventDisableTiming) != hipSuccess) {
      create_error() = "lane stream / event creation failed";
      ldm_destroy(h);
      return -2;
    }
  }
  *out = h;
  return 0;
}

extern "C" void ldm_destroy(ldm_handle* h) {
  if (!h) return;
  DeviceGuard guard(h->device);
  (void)hipDeviceSynchronize();
  h->drain_profile();
  for (auto& g : h->graphs) g.destroy();
  for (auto& kv : h->raw)
    if (kv.second.d) (void)hipFree(kv.second.d);
  for (void* p : h->owned) (void)hipFree(p);
  for (void* p : h->derived) (void)hipFree(p);
  if (h->loop_a) (void)hipEventDestroy(h->loop_a);
  if (h->loop_b) (void)hipEventDestroy(h->loop_b);
  if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
  for (auto st : h->lane_stream)
    if (st) (void)hipStreamDestroy(st);
  for (auto ev : h->lane_done)
    if (ev) (void)hipEventDestroy(ev);
  delete h;
}

// ------------------------------------------------------------------------------------------ parity hooks
extern "C" int ldm_denoise_logits(ldm_handle* h, const int32_t* d_tokens, int t, int B, float* d_logits, void* stream) {
  int rc = check_ready(h, B);
  if (rc) return rc;
  if (!d_tokens || !d_logits) return h->fail(-1, "null argument");
  if (t < 0 || t >= h->T) return h->fail(-1, "timestep out of range");
  ON_DEVICE(h);
  hipStream_t st = (hipStream_t)stream;
  for (int off = 0; off < B; off += h->chunk) {
    const int Bc = std::min(h->chunk, B - off);
    if ((rc = denoise_chunk(h, d_tokens + (size_t)off * h->S, t, Bc, st))) return rc;
    HIP_OK(h, hipMemcpy2DAsync(d_logits + (size_t)off * h->S * h->C, (size_t)h->C * 4, h->logits, (size_t)h->Cp * 4,
                               (size_t)h->C * 4, (size_t)Bc * h->S, hipMemcpyDeviceToDevice, st));
  }
  HIP_OK(h, hipGetLastError());
  return 0;
}

extern "C" int ldm_posterior(ldm_handle* h, const float* d_logits, const int32_t* d_tokens, int t_post, int B,
                             const ldm_cond* cond, float* d_logp, void* stream) {
  int rc = check_ready(h, B);
  if (rc) return rc;
  if (!d_logits || !d_tokens || !d_logp) return h->fail(-1, "null argument");
  if (t_post < 0 || t_post >= h->T) return h->fail(-1, "timestep out of range");
  ON_DEVICE(h);
  PostArgs p{};
  fill_post(h, p, cond, nullptr, 0, B);
  p.logits = d_logits;
  p.ldl = h->C;
  p.tokens = d_tokens;
  p.logp_out = d_logp;
  p.t_post = t_post;
  launch_posterior_sample(p, (hipStream_t)stream);
  HIP_OK(h, hipGetLastError());
  return 0;
}

int ldm_host::set_rng(ldm_handle* h, uint64_t seed, uint64_t first_layout, hipStream_t st) {
  launch_set_rng(h->rng, seed, first_layout, st);  // kernel args are captured by value at launch
  return 0;
}

extern "C" int ldm_sample_tokens(ldm_handle* h, const float* d_logp, const ldm_cond* cond, const ldm_sampler* s,
                                 uint64_t seed, uint64_t first_layout, int step, int B, int32_t* d_tokens_out,
                                 void* stream) {
  int rc = check_ready(h, B);
  if (rc) return rc;
  if ((rc = check_sampler(h, s))) return rc;
  if (!d_logp || !d_tokens_out) return h->fail(-1, "null argument");
  ON_DEVICE(h);
  hipStream_t st = (hipStream_t)stream;
  if ((rc = set_rng(h, seed, first_layout, st))) return rc;
  PostArgs p{};
  fill_post(h, p, nullptr, s, 0, B);
  if (cond) {  // only the [PAD] disabling applies at this stage (base.py:272-284)
    p.cond_seq = cond->d_cond_seq;
    p.pad_disable = cond->pad_disable;
  }
  p.logp_in = d_logp;
  p.tokens_out = d_tokens_out;
  p.step = step;
  launch_posterior_sample(p, st);
  HIP_OK(h, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------ relation
extern "C" int ldm_relation_update(ldm_handle* h, float* d_logp_inout, const int32_t* d_cond_seq,
                                   const ldm_relation* rel, int t, int B, void* stream) {
  if (!h) return -1;
  if (B <= 0) return B == 0 ? 0 : h->fail(-1, "negative batch");
  if (!d_logp_inout || !d_cond_seq || !rel) return h->fail(-1, "null argument");
  if (!rel->d_edge_offsets || !rel->d_centres) return h->fail(-1, "ldm_relation: null edge offsets / centres");
  if (rel->n_graph_total < B) return h->fail(-1, "ldm_relation.n_graph_total smaller than B");
  if (h->cfg.max_elem > 32 || h->cfg.n_bin > 32) return h->fail(-4, "relation kernel: max_elem and n_bin must be <= 32");
  for (int x = 0; x < 4; ++x)
    if (rel->canvas_bins[x] < 0 || rel->canvas_bins[x] >= h->cfg.n_bin) return h->fail(-1, "canvas bin out of range");
  if (t < 10 || rel->num_update <= 0) return 0;  // logit_adjustment.py:107
  ON_DEVICE(h);
  RelArgs a{};
  a.logp = d_logp_inout;
  a.cond_seq = d_cond_seq;
  fill_rel(h, a, rel, 0, B);
  launch_relation_update(a, (hipStream_t)stream);
  HIP_OK(h, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------ decode
extern "C" int ldm_decode_layouts(ldm_handle* h, const int32_t* d_tokens, int B, const double* d_centres, int box_f64,
                                  void* d_bbox, int64_t* d_label, uint8_t* d_mask, void* stream) {
  if (!h) return -1;
  if (B < 0) return h->fail(-1, "negative batch");
  if (B == 0) return 0;
  if (!d_tokens || !d_bbox || !d_label || !d_mask) return h->fail(-1, "null argument");
  ON_DEVICE(h);  // (n_attr == 5, i.e. c-x-y-w-h, is enforced by ldm_create)
  launch_decode_layouts(d_tokens, B, h->cfg.max_elem, h->cfg.n_attr, h->cfg.n_category, h->cfg.n_bin, d_centres,
                        box_f64, d_bbox, d_label, d_mask, (hipStream_t)stream);
  HIP_OK(h, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------ near-tie report
// Deterministic decoding in the fp16 mode is bit-exact against the reference wherever the winning class leads the
// runner-up by more than the mode's logits error can move.  With the report enabled every deterministic step marks
// the layouts in which some token was decided inside that band; the caller re-decides exactly those in the exact
// mode (layout_dm_amd.verified: greedy decoding is RNG-free and layouts are independent).
extern "C" int ldm_set_tie_report(ldm_handle* h, float tie_rel, float tie_abs) {
  if (!h) return -1;
  if (!(tie_rel >= 0.f) || !(tie_abs >= 0.f)) return h->fail(-1, "tie_rel / tie_abs must be >= 0");
  if (tie_abs > 0.f && !(tie_rel > 0.f)) tie_rel = 1e-30f;  // (the report is keyed on tie_rel > 0)
  ON_DEVICE(h);
  if (tie_rel > 0.f && !h->tie_flags) {
    h->tie_steps = std::max(h->T, 1);
    int rc = h->dalloc(&h->tie_flags, (size_t)h->tie_steps * h->cfg.max_batch);
    if (rc) return rc;
  }
  if ((tie_rel != h->tie_rel || tie_abs != h->tie_abs) && !h->graphs.empty()) {
    // captured graphs carry the flag pointers / thresholds of their capture; a replay on another stream may still be
    // executing one of them
    HIP_OK(h, hipDeviceSynchronize());
    for (auto& g : h->graphs) g.destroy();
    h->graphs.clear();
  }
  h->tie_rel = tie_rel;
  h->tie_abs = tie_abs;
  return 0;
}
extern "C" int ldm_get_tie_flags(ldm_handle* h, uint8_t* d_flags, int n_steps, int B, void* stream) {
  if (!h || !d_flags) return h ? h->fail(-1, "null argument") : -1;
  if (!h->tie_flags) return h->fail(-1, "near-tie report not enabled (ldm_set_tie_report)");
  if (n_steps < 1 || n_steps > h->tie_steps || B < 1 || B > h->cfg.max_batch) return h->fail(-1, "n_steps / B out of range");
  ON_DEVICE(h);
  HIP_OK(h, hipMemcpy2DAsync(d_flags, (size_t)B, h->tie_flags, (size_t)h->cfg.max_batch, (size_t)B, (size_t)n_steps,
                             hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
// ------------------------------------------------------------------------------------------ introspection
// "key=value;..." description of what this handle runs: numerics mode, kernel family, chunk / lanes, near-tie thresholds,
// the batch-shape rule of the one-launch loop (one workgroup per layout on the chip's compute units: a call costs whole
// rounds of `round` layouts, so B = round + 1 costs what B = 2 * round costs) and the development knobs the library has
// honoured in this process (ldm_knobs.h; always the LAST key: its value may itself contain ';').  Returns the length
// needed (excluding the terminator), whatever `cap` is: call again with a larger buffer when the result is >= cap.
#ifndef LDM_SRC_DIGEST
#define LDM_SRC_DIGEST "unknown"
#endif
// sha256 over the sources this library was built from (layout_dm_amd/build.py source_digest(), passed at compile time): the
// marker is what build.py / the tests scan a built .so for — a pushed tree whose prebuilt library lags its sources is rebuilt
// (or refused), not silently used
extern "C" const char ldm_build_source_digest[] = "LDM_SRC_DIGEST=" LDM_SRC_DIGEST;

extern "C" int ldm_describe(const ldm_handle* h, char* buf, int cap) {
  if (!h) return -1;
  static const char* prec[5] = {"exact_f32", "fast_f16", "split_f16", "mixed_f16", "hybrid_f16"};
  const bool loop = loop_fusable(h, nullptr);
  char num[96];
  std::string s = "abi=" + std::to_string(LDM_ABI_VERSION) + ";precision=" + prec[h->mixed ? 2 + h->mixed : h->cfg.precision];
  std::string kern = h->fused_attn == 6 ? "stack" : "generic16";
  if (h->cfg.precision != LDM_PREC_FAST_F16) {
    kern = "tiled_gemm+attn";
    if (h->lngemm)
      kern = std::string("row_resident_ln_gemm") + (h->attn_ffn_fused ? "+attn_ffn_fused_fp16" : h->ffn_fused ? "+ffn_fused_fp16" : h->pre_ffn2 ? "+linear2_prologue" : "") + (h->pre_out ? "+out_proj_prologue" : "") +
             (h->attnout ? "+attn_out_proj_fused" : h->pre_ffn2 && h->pre_out ? "+attn" : "+tiled_gemm+attn");
  }
  s += ";kernels=" + kern;
  s += std::string(";loop=") + (loop ? "one_launch" : "per_step_graph");
  s += ";chunk=" + std::to_string(h->chunk) + ";lanes=" + std::to_string(h->n_lanes) + ";lane_offset_us=" + std::to_string(h->lane_offset_us);
  snprintf(num, sizeof(num), ";tie_rel=%g;tie_abs=%g", (double)h->tie_rel, (double)h->tie_abs);
  s += num;
  // batch quantum: the loop kernel runs one workgroup per layout, one workgroup per compute unit; the per-step path runs
  // `chunk` layouts per pass on `lanes` concurrent pipelines
  s += ";round=" + std::to_string(loop ? h->n_cu : h->chunk * std::max(h->n_lanes, 1));
  s += std::string(";batch_rule=") + (loop ? "whole_rounds_of_one_workgroup_per_layout" : "whole_chunks");
  s += std::string(";src_digest=") + (ldm_build_source_digest + sizeof("LDM_SRC_DIGEST=") - 1);
  s += ";knobs=" + knobs_honoured();
  if (buf && cap > 0) {
    const size_t n = std::min((size_t)cap - 1, s.size());
    memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return (int)s.size();
}

extern "C" int ldm_last_loop_ms(ldm_handle* h, float* ms) {
  if (!h || !ms) return -1;
  if (!h->loop_timed) return h->fail(-1, "no loop has run yet");
  HIP_OK(h, hipEventSynchronize(h->loop_b));
  HIP_OK(h, hipEventElapsedTime(ms, h->loop_a, h->loop_b));
  return 0;
}

extern "C" int ldm_set_profiling(ldm_handle* h, int enable) {
  if (!h) return -1;
  h->drain_profile();
  h->profiling = enable != 0;
  return 0;
}
extern "C" int ldm_profile_count(ldm_handle* h) {
  if (!h) return -1;
  h->drain_profile();
  return (int)h->prof.size();
}
extern "C" int ldm_profile_get(ldm_handle* h, int idx, const char** name, double* total_ms, int64_t* launches,
                               double* flops, double* bytes) {
  if (!h || idx < 0 || idx >= (int)h->prof.size()) return -1;
  h->drain_profile();
  const ProfEntry& e = h->prof[idx];
  if (name) *name = e.name.c_str();
  if (total_ms) *total_ms = e.ms;
  if (launches) *launches = e.launches;
  if (flops) *flops = e.flops;
  if (bytes) *bytes = e.bytes;
  return 0;
}
extern "C" int ldm_profile_reset(ldm_handle* h) {
  if (!h) return -1;
  h->drain_profile();
  h->prof.clear();
  return 0;
}

