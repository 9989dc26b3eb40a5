/* ldm_hip.h — C-ABI of the MI355X-native LayoutDM sampling hot path (libldm_hip.so).
 *
 * The reference (CyberAgentAILab/layout-dm) is 100 % Python/PyTorch and has no FFI layer;
 * its narrowest seam that contains exactly the hot path is the Python method
 *   ConstrainedMaskAndReplaceDiffusion.sample(batch_size, cond, sampling_cfg, ...)
 *     src/trainer/trainer/models/categorical_diffusion/base.py:293-371
 * reached from LayoutDM.sample (models/layoutdm.py:77-88) and test.py:195-200.
 * These entry points are what a ctypes binding placed at that seam calls (the binding is
 * shown in INTEGRATION.md and implemented in layout_dm_amd/binding.py).
 *
 * Conventions: every pointer named d_* is a DEVICE pointer owned by the caller (e.g. a torch
 * tensor's data_ptr()); h_* are host pointers.  `stream` is a hipStream_t passed as void*.
 * Return value 0 = OK, negative = error (text via ldm_last_error).  A handle is bound to one
 * device and is not thread-safe.  No entry point synchronises the device except
 * ldm_create / ldm_finalize_weights / ldm_destroy and the *_sync helpers.
 */
#ifndef LDM_HIP_H
#define LDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDM_ABI_VERSION 5

typedef struct ldm_handle ldm_handle;

/* numerics mode of the denoiser GEMMs / attention */
enum {
  LDM_PREC_EXACT_F32 = 0, /* v_mfma_f32_32x32x2_f32: exact fp32 (== fmaf chain) */
  LDM_PREC_FAST_F16 = 1,  /* fp16 operands, fp32 accumulate (v_mfma_f32_32x32x16_f16) */
  LDM_PREC_SPLIT_F16 = 2, /* fp16 hi+lo split operands, 3 MFMA passes, ~fp32 accuracy */
  LDM_PREC_MIXED_F16 = 3, /* the split mode with fp16-ONLY weights: activations (and the attention's q, k, v, P) stay hi+lo, the four
                             weight GEMMs and the head drop the W_lo product — 2 MFMA passes there, 3 inside the attention.  Logits
                             error 1e-4 .. 6e-4 of max |logit| on checkpoints whose fp16 error is ~1e-3;
                             reference backbone geometry only (ldm_create fails otherwise) */
  LDM_PREC_HYBRID_F16 = 4 /* mixed with the FFN and the vocabulary head in PLAIN fp16 (LayerNorm output, hidden activations and weights
                             rounded once: one MFMA pass there); the attention path — AdaLN output into in_proj, q, k, v, P, the attention
                             output into out_proj — keeps hi+lo activations.  Logits error 2.3e-4 on a trained checkpoint whose fp16 error
                             is 1.2e-3 (the fp16 error lives on the attention-score path: DESIGN.md section 3.5); same geometry rule */
};

/* transition-matrix family == Q_TYPES of the reference (models/layoutdm.py:20-23) */
enum {
  LDM_Q_CONSTRAINED = 0, /* ConstrainedMaskAndReplaceDiffusion: one sub-vocabulary per attribute
                            (categorical_diffusion/constrained.py) — the LayoutDM default */
  LDM_Q_VANILLA = 1      /* VanillaMaskAndReplaceDiffusion: one vocabulary of all C classes, MASK last,
                            un-prefixed schedule buffers (categorical_diffusion/vanilla.py) */
};

/* sampler kinds: trainer/helpers/sampling.py:13-59,81-130 */
enum {
  LDM_SAMPLE_DETERMINISTIC = 0,
  LDM_SAMPLE_RANDOM = 1,
  LDM_SAMPLE_TOP_P = 2,
  LDM_SAMPLE_TOP_K = 3,
  LDM_SAMPLE_GUMBEL = 4
};

/* Model geometry.  Mirrors what the reference derives from its hydra config:
 * vocabulary (helpers/layout_tokenizer.py:79-82,152-153), backbone
 * (config/backbone/medium.yaml shrunk 29/32 at models/layoutdm.py:54), T (layoutdm.py:33). */
typedef struct {
  int32_t abi_version; /* LDM_ABI_VERSION */
  int32_t n_category;  /* 25 (Rico25) / 5 (PubLayNet) */
  int32_t n_bin;       /* bins per coordinate (32) */
  int32_t max_elem;    /* 25 */
  int32_t n_attr;      /* 5 : c,x,y,w,h */
  int32_t d_model;     /* 464 */
  int32_t n_head;      /* 8 */
  int32_t d_ff;        /* 1856 */
  int32_t n_layer;     /* 4 */
  int32_t n_step;      /* T = 100 */
  int32_t precision;   /* LDM_PREC_* */
  int32_t max_batch;   /* largest B any call will use (workspace is sized for it) */
  int32_t chunk;       /* layouts processed per pass through the network (0 = auto = 256) so that the
                          activation working set stays inside the 256 MiB Infinity Cache */
  int32_t q_type;      /* LDM_Q_* (ABI 2) */
  int32_t lanes;       /* chunk pipelines run concurrently, each on its own stream / hipGraph, phase-shifted so that
                          their HBM-bound phases do not coincide (0 = auto = 2, 1 = one pipeline; ABI 3) */
} ldm_config;

/* sampling_cfg of the reference (helpers/sampling.py dataclasses) */
typedef struct {
  int32_t kind;      /* LDM_SAMPLE_* */
  float temperature; /* logits / temperature (stochastic kinds only) */
  float top_p;       /* LDM_SAMPLE_TOP_P */
  int32_t top_k;     /* LDM_SAMPLE_TOP_K */
} ldm_sampler;

/* Optional constraints == the `cond` dict consumed at base.py:243-284. All may be NULL/0. */
typedef struct {
  const int32_t* d_cond_seq;    /* (B,S) cond["seq"] */
  const uint8_t* d_strong_mask; /* (B,S) cond["mask"]: 1 = token fixed to cond_seq (base.py:245-251) */
  const float* d_weak_logits;   /* (B,C,S) refinement prior, added where !strong (base.py:254-258) */
  int32_t pad_disable;          /* cond["type"] in {c,cwh,refinement,relation} (base.py:272-284) */
} ldm_cond;

/* cond=relation: the graph of cond["batch_w_canvas"] (helpers/task.py:112-114) re-indexed per layout — node 0 =
 * canvas, node k = k-th element whose conditioned category is not PAD — plus the hyper-parameters of the logit
 * adjustment the reference applies between the posterior and the draw (base.py:261-269 -> update(),
 * categorical_diffusion/logit_adjustment.py:88-126, losses models/clg/const.py:221-236), relation_mode = "average". */
typedef struct {
  const int32_t* d_edge_offsets; /* (B+1) CSR offsets into the edge arrays */
  const int32_t* d_edge_src;     /* (n_edges) node id inside its layout's graph */
  const int32_t* d_edge_dst;
  const int32_t* d_edge_attr;    /* 1<<RelSize | 1<<RelLoc bitmasks (trainer/data/util.py:14-27,168) */
  const float* d_centres;        /* (4, n_bin) cluster centres in x,y,w,h order (float32, as update() casts them) */
  int32_t canvas_bins[4];        /* bbox_tokenizer.encode([0.5,0.5,1,1]) per coordinate, 0..n_bin-1 */
  float relation_lambda;         /* sampling_cfg.relation_lambda (SGD learning rate) */
  int32_t num_update;            /* sampling_cfg.relation_num_update */
  int32_t n_graph_total;         /* batch size of the whole sampling call: the loss is a mean over 14*B terms */
} ldm_relation;

/* ---- lifecycle ----------------------------------------------------------------------
 * ldm_create validates the whole geometry BEFORE it touches a device and returns -1 with ldm_last_error(NULL) naming
 * the field: n_attr == 5; d_model, d_ff multiples of 16, d_model <= 1024; d_model % n_head == 0, head dimension <= 64;
 * n_category + 4 n_bin + 2 <= 192 classes; max_elem * n_attr <= 128 tokens in LDM_PREC_FAST_F16 (one score tile per
 * head; longer sequences: LDM_PREC_EXACT_F32, bounded by the LDS: 2 * S * head_dim floats <= 160 KiB).  The reference's
 * own configurations (rico25 / publaynet on the 464 / 8 / 1856 backbone, S = 125) run on the layout-resident kernels,
 * every other accepted geometry on generic tiled kernels (same numerics contract, lower throughput). */
int ldm_create(const ldm_config* cfg, int device, ldm_handle** out);
void ldm_destroy(ldm_handle* h);
const char* ldm_last_error(const ldm_handle* h); /* h may be NULL: last create error */

/* ---- weights: the reference checkpoint format ------------------------
 * key = reference state_dict key with or without the "model.module." prefix; data = host
 * float32, C-contiguous, shape as in the checkpoint.  Replaces model.load_state_dict
 * (models/common/util.py:47-57, test.py:144-149).  The handle owns repacked device copies. */
int ldm_load_weight(ldm_handle* h, const char* key, const float* h_data, const int64_t* shape, int ndim);
/* builds the (S,D) positional table (nn_lib.py:112-127), the [T][L][2D] AdaLN table
 * (transformer_utils.py:79-81), fp16 weight copies; fails if a required key is missing. */
int ldm_finalize_weights(ldm_handle* h);

/* ---- parity hooks (one stage each) --------------------------------------------------- */
/* CategoricalTransformer.forward (nn_lib.py:191-237): tokens (B,S) -> logits (B,S,C) fp32 */
int ldm_denoise_logits(ldm_handle* h, const int32_t* d_tokens, int t, int B, float* d_logits, void* stream);
/* predict_start tail + q_posterior + cond overrides (base.py:131-144, constrained.py:135-206,
 * base.py:243-284): logits (B,S,C), tokens (B,S) -> log p(x_{t-1}|x_t) (B,C,S) fp32.
 * t_post = the timestep handed to q_posterior (noise_t [- skip_step], base.py:218-240). */
int ldm_posterior(ldm_handle* h, const float* d_logits, const int32_t* d_tokens, int t_post, int B,
                  const ldm_cond* cond, float* d_logp, void* stream);
/* helpers/sampling.py:81-130 on a (B,C,S) log-prob tensor -> (B,S) int32 tokens.  cond (may be NULL): only
 * d_cond_seq + pad_disable are used — the [PAD] disabling of base.py:272-284, which the reference applies right
 * before the draw (for cond=relation: after the logit adjustment). */
int ldm_sample_tokens(ldm_handle* h, const float* d_logp, const ldm_cond* cond, const ldm_sampler* s, uint64_t seed,
                      uint64_t first_layout, int step, int B, int32_t* d_tokens_out, void* stream);

/* ---- the hot path -------------------------------------------------------------------- */
/* One reverse step (_sample_single_step, base.py:205-291), fused: tokens (B,S) -> tokens.
 * rel (may be NULL): cond["type"] == "relation" — the step then runs posterior (+ strong mask) -> logit adjustment
 * (t_model >= 10) -> [PAD] disable -> draw, exactly the reference's order. */
int ldm_sample_step(ldm_handle* h, const int32_t* d_tokens_in, int32_t* d_tokens_out, int t_model,
                    int t_post, const ldm_cond* cond, const ldm_relation* rel, const ldm_sampler* s, uint64_t seed,
                    uint64_t first_layout, int step, int B, void* stream);
/* The T-step reverse loop (BaseMaskAndReplaceDiffusion.sample, base.py:293-371).
 * d_tokens_inout: initial state (all [MASK] for unconditional, cond["seq"] otherwise) -> final.
 * h_t_model / h_t_post: host arrays of n_steps timesteps (diffusion_list and the posterior's t).
 * rel: NULL, or the relation graph of the B layouts (cond must then carry d_cond_seq).
 * d_intermediates: optional (n_steps,B,S) int32 (get_intermediate_results=True).
 * In LDM_PREC_FAST_F16 on the reference's backbone (and without rel) the whole loop is ONE launch: every layout's
 * workgroup runs all its steps (tokens in LDS, the step's tail behind the vocabulary head); use_graph is then moot.
 * Otherwise, use_graph != 0: the whole loop is captured once per (B, schedule, sampler, cond / relation layout) into
 * a hipGraph and replayed (seed / first_layout live in device memory so replays may change them; cond tensors,
 * the relation graph and the intermediates are staged through handle-owned buffers at fixed addresses). */
int ldm_sample_loop(ldm_handle* h, int32_t* d_tokens_inout, const ldm_cond* cond, const ldm_relation* rel,
                    const int32_t* h_t_model, const int32_t* h_t_post, int n_steps, const ldm_sampler* s,
                    uint64_t seed, uint64_t first_layout, int B, int32_t* d_intermediates, int use_graph,
                    void* stream);

/* ---- near-tie report of deterministic decoding (ABI 4; tie_abs: ABI 5) ------------------------
 * north star: "token indices bit-exact under greedy/argmax decoding".  LDM_PREC_FAST_F16's logits carry an error that
 * depends on the checkpoint (3e-4 of max |logit| on the reference's init, 1e-3 on wider weights, more once attention rows
 * saturate: DESIGN.md section 3.5), so its argmax (sampling.py:88-90) can differ from the reference's where two classes
 * are closer than that error can move them.  With the report enabled every deterministic ldm_sample_step /
 * ldm_sample_loop marks, per (step, layout), whether some token of the layout was decided with a log-probability lead
 * over the runner-up below  max(tie_rel * max |logit of that token|, tie_abs).  The lead moves by at most 6 x the largest
 * absolute logits error of the token: with tie_abs = 6 x a bound on that error — MEASURED on the
 * checkpoint by the caller, layout_dm_amd/verified.py calibrate() — an unmarked token is the exact mode's token; the
 * caller re-checks the marked (step, layout) pairs in LDM_PREC_EXACT_F32.  tie_rel = tie_abs = 0 disables.  Not defined
 * for cond=relation (the draw follows an SGD on the log-probabilities): such calls fail while the report is enabled. */
int ldm_set_tie_report(ldm_handle* h, float tie_rel, float tie_abs);
/* flags of the most recent deterministic call: d_flags (n_steps, B) uint8, row i = i-th step of that call */
int ldm_get_tie_flags(ldm_handle* h, uint8_t* d_flags, int n_steps, int B, void* stream);

/* ---- cond=relation, split-step form ------------------------------------------------------- */
/* The logit adjustment alone (`num_update` SGD steps on the mean relational-constraint loss, analytic gradient;
 * no-op for t < 10, logit_adjustment.py:107), for callers that drive the step stage by stage: call between
 * ldm_posterior (cond without pad_disable) and ldm_sample_tokens (cond with pad_disable). */
int ldm_relation_update(ldm_handle* h, float* d_logp_inout, const int32_t* d_cond_seq, const ldm_relation* rel,
                        int t, int B, void* stream);

/* ---- result packaging ----------------------------------------------------------------- */
/* ids -> {bbox, label, mask}: LayoutSequenceTokenizer.decode (helpers/layout_tokenizer.py:255-266) +
 * BboxTokenizer.decode (helpers/bbox_tokenizer.py:117-168) for var_order c-x-y-w-h with the stacked
 * x-y-w-h bbox vocabulary and no bos/eos (the LayoutDM configuration), which LayoutDM.sample
 * (models/layoutdm.py:77-88) runs on the host inside the reference's timed region (test.py:194-203).
 * d_tokens: (B,S) int32.  d_centres: NULL for bbox_quantization=linear, else the (4,n_bin) float64
 * cluster centres in x,y,w,h order (kmeans / percentile).  box_f64: 0 -> d_bbox is (B,E,4) float32
 * (what the reference returns for linear), 1 -> float64 (what it returns for kmeans/percentile).
 * d_label: (B,E) int64, d_mask: (B,E) uint8 (bool). */
int ldm_decode_layouts(ldm_handle* h, const int32_t* d_tokens, int B, const double* d_centres, int box_f64,
                       void* d_bbox, int64_t* d_label, uint8_t* d_mask, void* stream);

/* ---- FID feature extractor ------------------------------------------------
 * FIDNetV3.extract_features (trainer/fid/model.py:123-164): {bbox, label, padding_mask} -> the 256-d feature of the
 * [token] slot, which trainer/eval.py feeds to compute_generative_model_scores (helpers/metric.py:37-59: FID,
 * precision / recall / density / coverage).  Own handle: the network has its own checkpoint (model.py:182-193).
 * d_bbox (B,N,4) float32, d_label (B,N) int64, d_padding_mask (B,N) uint8 (1 = padded, i.e. ~mask), d_feat (B,256). */
typedef struct ldm_fid ldm_fid;
int ldm_fid_create(int num_label, int max_bbox, int d_model, int n_head, int n_layer, int device, ldm_fid** out);
void ldm_fid_destroy(ldm_fid* h);
const char* ldm_fid_last_error(const ldm_fid* h); /* h may be NULL: last create error */
/* key = FIDNetV3 state_dict key (emb_label.weight, fc_bbox.*, enc_fc_in.*, enc_transformer.token,
 * enc_transformer.core.layers.<i>.{self_attn.in_proj_*, self_attn.out_proj.*, linear1.*, linear2.*, norm1.*, norm2.*});
 * decoder-half keys are accepted and ignored.  host float32, C-contiguous. */
int ldm_fid_load_weight(ldm_fid* h, const char* key, const float* h_data, const int64_t* shape, int ndim);
int ldm_fid_finalize(ldm_fid* h);
int ldm_fid_features(ldm_fid* h, const float* d_bbox, const int64_t* d_label, const uint8_t* d_padding_mask, int B,
                     int N, float* d_feat, void* stream);

/* Precision / recall / density / coverage of two feature sets (the other four entries of compute_generative_model_scores,
 * helpers/metric.py:37-59, which the reference takes from prdc.compute_prdc(real_features, fake_features, nearest_k=5)):
 * d_real (n_real, dim), d_fake (n_fake, dim) float32 device; h_out4 = {precision, recall, density, coverage} on the HOST
 * (the call synchronises `stream`); 1 <= nearest_k <= 7.  Uses the current device; workspace is allocated and freed
 * inside the call (max(n_real, n_fake)^2 floats; at most 65 536 features per set, -6 beyond). */
int ldm_prdc(const float* d_real, int n_real, const float* d_fake, int n_fake, int dim, int nearest_k, float* h_out4,
             void* stream);

/* ---- alignment / overlap of generated layouts (r05; the remaining per-layout metrics of eval.py) ---------------------
 * compute_alignment + compute_overlap (trainer/helpers/metric.py:98-203, called on every generated batch at
 * eval.py:153-155,203-205) on decoded layouts resident in HBM — what ldm_decode_layouts wrote: d_bbox (B,S,4) float32
 * (xc, yc, w, h), d_mask (B,S) uint8 (1 = valid element), 1 <= S <= 256.  d_out6 (B,6) float32, per layout:
 * alignment-ACLayoutGAN, alignment-LayoutGAN++, alignment-NDN, overlap-ACLayoutGAN, overlap-LayoutGAN++,
 * overlap-LayoutGAN (the reference's dictionary keys, in its order).  fp32 like the reference; sums in index order
 * (parity to fp32 rounding).  Uses the current device; no handle.  Returns 0, -1 (bad argument) or -2 (launch failed). */
int ldm_layout_metrics(const float* d_bbox, const uint8_t* d_mask, int B, int S, float* d_out6, void* stream);

/* ---- introspection ------------------------------------------------------------------- */
/* average device time (ms) of the most recent ldm_sample_loop, measured with HIP events on the
 * stream it ran on; blocks until that loop has finished. */
int ldm_last_loop_ms(ldm_handle* h, float* ms);
/* name + accumulated ms + launches of every kernel class timed when profiling is enabled
 * (ldm_set_profiling(h,1) brackets each launch with HIP events; slow, for bench/roofline only) */
int ldm_set_profiling(ldm_handle* h, int enable);
int ldm_profile_count(ldm_handle* h);
int ldm_profile_get(ldm_handle* h, int idx, const char** name, double* total_ms, int64_t* launches,
                    double* flops, double* bytes);
int ldm_profile_reset(ldm_handle* h);
int ldm_abi_version(void);
/* "key=value;..." description of what the handle runs — numerics mode, kernel family, one-launch loop or per-step
 * graphs, chunk / lanes, near-tie thresholds — and the development knobs the library honoured in this process.
 * Environment knobs (LDM_STACK_LOOP, LDM_FUSED_ATTN, ...: INTEGRATION.md section 5) select development / ablation paths;
 * they are honoured only together with LDM_DEV=1, and ldm_create fails while one is set without it.
 * Writes at most cap - 1 characters + NUL; returns the length needed (ABI 5). */
int ldm_describe(const ldm_handle* h, char* buf, int cap);

/* Build provenance: "LDM_SRC_DIGEST=" followed by the sha256 (64 hex digits) of the sources this library was built from
 * (layout_dm_amd/build.py source_digest(): csrc, this header, the source list and the flags).  build.py and binding.py compare
 * it with the tree before loading a prebuilt library; ldm_describe reports it as src_digest. */
extern const char ldm_build_source_digest[];
/* layouts per chunk and number of concurrent lanes the handle was created with (after the 0 = auto defaults) */
int ldm_get_layout(const ldm_handle* h, int* chunk, int* lanes);

#ifdef __cplusplus
}
#endif
#endif /* LDM_HIP_H */
