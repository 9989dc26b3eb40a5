// Internal launch interface between the C-ABI host layer (ldm_api.cpp, ldm_loop.cpp) and the gfx950 kernels.
// Everything here is CDNA4-only (wave64, MFMA); there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <cstdio>
#include <mutex>
#include <set>
#include <utility>

#include "ldm_knobs.h"

namespace ldm {

// Kernels that use more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once
// per (device, kernel): the attribute is per device, so a process-wide "done" flag is wrong as soon as a second
// handle lives on another GPU.  Thread-safe; a failure is reported once on stderr (the launch then fails loudly).
inline void allow_big_lds(const void* kern) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  if (!done.insert({dev, kern}).second) return;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    fprintf(stderr, "libldm_hip: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed on device %d: %s\n", dev,
            hipGetErrorString(e));
    (void)hipGetLastError();  // the launch that follows reports its own status
  }
}

constexpr float kLogEps = -69.07755278982137f;  // log(1e-30): categorical_diffusion/util.py:8
// Split mode (LDM_PREC_SPLIT_F16): x = hi + lo / kSplitLoScale with hi = fp16(x), lo = fp16((x - hi) * kSplitLoScale).
// r04: 1 (was 2^11).  An unscaled lo lets the three products A_hi W_hi + A_lo W_hi + A_hi W_lo share ONE accumulator (the
// 2^11 convention needed a second one for the cross terms: half the accumulator registers, i.e. half the tile).  fp16 keeps
// denormals, so lo carries an absolute error <= 2^-25: negligible for operands of magnitude ~1 — LayerNorm outputs,
// attention outputs, the FFN's hidden activations — and the WEIGHTS are brought there by an exact power-of-two pre-scale
// per tensor (ldm_weights.cpp make_w16), undone in the GEMM's epilogue (GemmArgs.out_scale).  Logits error against the
// float64 restatement, CPU emulation on three weight distributions: 6.0e-7 / 1.1e-6 / 1.0e-4 — the fp32 path's own.
constexpr float kSplitLoScale = 1.0f;
constexpr int kMaxAttr = 8;

// ---- row kernels (kernels_norm.hip) -----------------------------------------------------
// y = LN(x) * (1 + scale) + shift        (AdaLayerNorm, transformer_utils.py:79-83)   ada = 1
// y = LN(x) * gamma + beta               (nn.LayerNorm)                               ada = 0
// x = emb[token] + pos[s] first if tokens != nullptr (nn_lib.py:204,220)
struct LnArgs {
  const float* x;         // [M, D] (ignored when tokens != nullptr)
  const int32_t* tokens;  // [M] or nullptr
  const float* emb;       // [C, D]
  const float* pos;       // [S, D]
  const float* p0;        // scale (ada) or gamma   [D]
  const float* p1;        // shift (ada) or beta    [D]
  float* y32;             // [M, D] or nullptr
  __half* y16;            // [M, ld16] or nullptr   (GEMM A operand, fp16 modes)
  __half* y16lo;          // [M, ld16] or nullptr   (split mode: residual x - fp16(x), times kSplitLoScale)
  float2* stats_out;      // [M] (mean, rstd) or nullptr   (deferred normalisation, fast mode)
  int M, D, S, ld16, ada;
  int raw;                // 1: y32 receives the UN-normalised x (embedding only), stats_out the statistics
};
void launch_layernorm(const LnArgs& a, hipStream_t st);

// ---- GEMM (kernels_gemm.hip):  C[M,N] = epi(A[M,K] * W[N,K]^T + bias) ---------------------
struct GemmArgs {
  const void* A;      // fp32 (exact) or fp16 (fast/split) [M, lda]
  const void* Alo;    // split mode: low halves of A
  const void* W;      // [N, ldw] same dtype family as A
  const void* Wlo;    // split mode
  const float* bias;  // [N] or nullptr
  const float* res;   // residual [M, ldres] fp32 or nullptr
  float* C32;         // [M, ldc32] or nullptr
  __half* C16;        // [M, ldc16] or nullptr
  __half* C16lo;      // split mode, or nullptr
  int M, N, K, lda, ldw, ldres, ldc32, ldc16;
  int relu;
  int precision;  // LDM_PREC_*
  float out_scale;  // split mode: 2^-k of the weight tensor's pre-scale 2^k (0 = 1)
};
void launch_gemm(const GemmArgs& g, hipStream_t st);

// Row-resident (Ada)LayerNorm + fp16 x 3 GEMM (kernels_lngemm.hip): out = epi(LN(x) W^T) with the normalised hi / lo
// fragments of 128 rows resident in a workgroup's registers and the weights streamed as hi | lo tile images
// (ldm_pack::pack_x3_tile_image).  D == 464, n_tiles even, N % 4 == 0, N <= 32 n_tiles <= 2048.  -1: geometry not supported.
struct LnGemmArgs {
  const float* x;           // [M, ldx] rows (ignored when tokens != nullptr)
  const int32_t* tokens;    // [M] or nullptr: x = emb[token] + pos[row % S]
  const float *emb, *pos;
  const float *p0, *p1;     // ada: AdaLN scale / shift; else gamma / beta   [D]
  float* y32;               // ada: the normalised rows (the block's residual base) [M, D] or nullptr (may alias x)
  const char* img;          // n_tiles x 64 KiB: W hi tile | W lo tile (K axis in k-slot order, tile swizzle applied)
  const float* bias;        // [N] or nullptr
  float* C32;               // [M, ldc32] or nullptr
  __half *C16, *C16lo;      // [M, ldc16] hi / lo, or nullptr
  int M, N, D, S, ldx, ldc32, ldc16, n_tiles, ada, relu;
  float out_scale;          // 2^-k of the weight tensor's power-of-two pre-scale
  // r06: hi / lo fp16 output WITHOUT ReLU in PANEL-major form (in_proj in front of kernels_attnout.hip): C16 / C16lo are arrays
  // [N / 32 panels][panel_rows][32 halfs] — column c of row r at panel c / 32, byte r * 64 + (c % 32) * 2; ldc16 is ignored.
  // With relu = 1 (linear1): the same panel form for the hidden activations, read back by the linear2 GEMM prologue (pre_panel_stride)
  int panel_out;
  size_t panel_stride;      // bytes between panels (>= (M + slack) * 64, a multiple of 16)
  // GEMM prologue (pre_img != nullptr): x = pre_res + pre_bias + pre_scale * (preA · Wpre^T) is computed by the kernel itself instead
  // of being read — out_proj in front of norm2 + linear1, linear2 in front of the next AdaLN + in_proj / of the head
  const __half *preA, *preAlo;   // [M, pre_lda] hi / lo rows (attention output / hidden activations), K = 32 * pre_stages columns read
  const char* pre_img;           // pre_stages x 64 KiB K-slab image (ldm_pack::pack_x3_slab_image) of Wpre [D, K]
  const float* pre_bias;         // [D]
  const float* pre_res;          // [M, D] fp32 residual rows
  float* pre_out;                // [M, D] or nullptr: x written back (the residual base of a later GEMM)
  int pre_lda, pre_stages, pre_astages;   // stages of the image (a multiple of 3, zero slabs at the end) / stages with real K columns
  size_t pre_panel_stride;                // 0: preA / preAlo are row-major; else (r06) panel-major [K / 32][rows][32 halves], bytes between panels
  float pre_scale;
  // products per k16-step of the tile loop / of the GEMM prologue (0 = 3).  3: x = hi + lo on both operands (split).  2: WEIGHTS fp16 only —
  // W_hi x_hi + W_hi x_lo, the images' lo halves neither fetched nor read (LDM_PREC_MIXED_F16).  1: plain fp16 — the activation operand has no lo
  // half either (tile loop: the LayerNorm output is rounded once; prologue: preAlo is not read).  With C16lo == nullptr (ReLU, panel_out) the
  // epilogue writes plain-fp16 hidden panels (LDM_PREC_HYBRID_F16: the FFN and the head in plain fp16, the attention path in the two-product form)
  int np_main, np_pre;
};
int launch_lngemm16x3(const LnGemmArgs& a, hipStream_t st);
// Fused plain-fp16 FFN on rows (kernels_ffn16.hip, the hybrid mode): out = x + b2 + W2 relu(W1 LN(x; gamma, beta) + b1); x / out fp32 [M, D] (may alias),
// img = ldm_pack::pack_ffn_image_pipelined ((n_chunks + 1) x 64 KiB), D == 464, F == 32 n_chunks <= 2048.  -1: geometry not supported.
struct FfnRowsArgs {
  const float* x;
  float* out;
  const float *gamma, *beta, *b1, *b2;
  const char* img;
  int M, D, F, n_chunks;
};
int launch_ffn16_rows(const FfnRowsArgs& a, hipStream_t st);
void lngemm_phase_read(unsigned long long* out8);   // (LDM_LNGEMM_TM=1: accumulated phase cycles, reset on read)
// fp16 LDS-DMA pipelined GEMM (kernels_gemm16.hip); cfg selects the tile configuration
void launch_gemm16(const GemmArgs& g, int cfg, int tag, hipStream_t st);
int gemm16_block_k(int cfg);
// fp16 x 3 split GEMM on the same pipeline (kernels_gemm16.hip gemm16x3_k): A / Alo / W / Wlo, K % 32 == 0, operand
// buffers allocated to whole 128-row tiles
void launch_gemm16x3(const GemmArgs& g, int tag, hipStream_t st);
void launch_gemm16x3_abl(const GemmArgs& g, int abl, hipStream_t st);  // dev ablations (ldm_dev_bench_gemm_x3)
// per-layer weights of the stack kernel (all device pointers)
struct FusedLayerW {
  const void* img;        // ldm_pack::pack_attn_head_image (in_proj K axis in k-slot order)
  const float* bias_in;   // head-padded in_proj bias [3*H*64]
  const float *ada_scale, *ada_shift;  // AdaLN (scale, shift) of this layer at the step's timestep [D] each
  const float* b_out;     // out_proj bias [D]
  const void* ffn_img;    // pack_ffn_image, W1 K axis in k-slot order
  const float *b1, *b2, *g2, *be2;
};
struct FusedLayerSet {
  FusedLayerW w[8];
  int n_layer;
};
// the whole stack in one launch with the rows resident in the out-projection accumulators (kernels_stack.hip):
// ls.w[i].img = ldm_pack::pack_attn_head_image, .b_out = out_proj bias + W_out b_v
// head != nullptr: the vocabulary head (LayerNorm + Linear without bias) runs in the same workgroups and the kernel
// writes logits instead of rows (img: n_tiles x 32-KiB tile images, K axis in k-slot order: ldm_pack::pack_head_image)
struct PostArgs;
struct StackHead {
  const void* img;
  const float *g, *b;
  float* logits;
  int ldl, n_tiles;
};
void launch_stack_stream(const FusedLayerSet& ls, int F, float* x, int ldx, int B, int S, int H, int dh,
                         const StackHead& head, hipStream_t st);
// ... and the whole reverse loop of a layout in its workgroup (kernels_stack.hip HEAD == 2): tokens in / out through
// post->tokens / post->tokens_out, the step's tail (ldm_post_token.h) behind the vocabulary head on the logits in LDS.
// ls.w[i].ada_scale / ada_shift are ignored: the AdaLN rows of step i come from adaln[t_model[i]].
// parameter tables of the loop kernel as LDS images (floats; built on the host by ldm_weights.cpp build_loop_tables)
constexpr int kStackTblAttStatic = 1536;  // in_proj bias [3 * 8 heads * 64]
constexpr int kStackTblAttDyn = 1536;     // 1 + AdaLN scale [512] | shift [512] | b_out + W_out b_v + shift [512]
constexpr int kStackTblFfn = 3584;        // linear1 bias [2048] | norm2 gamma [512] | beta [512] | linear2 bias [512]
struct StackTables {
  const float* att_static;  // [L][kStackTblAttStatic]
  const float* att_dyn;     // [T][L][kStackTblAttDyn]
  const float* ffn;         // [L][kStackTblFfn]
  const float* head;        // [kStackTblAttDyn]: head LayerNorm gamma | beta | 0
};
struct StackLoop {
  StackTables tables;
  const PostArgs* post;      // schedule, cond, sampler, RNG, vocabulary, emb / pos / D; post->step = index of the first step
  const float* adaln;        // [T][L][2 N]
  const int32_t *t_model, *t_post;  // HOST arrays [n_steps], n_steps <= kStackLoopMaxSteps (longer loops: several launches)
  int32_t* inter;            // [n_steps][inter_ld][S] or nullptr
  int n_steps, inter_ld;
  int tie_ld;                // post->tie_flags of step i at + i * tie_ld (per layout of this launch)
  const struct RelArgs* rel; // cond=relation: the graph of the launch's layouts (edge_off advanced to its first layout), or nullptr
};
// largest sub-vocabulary (body + [PAD] + [MASK]) the fused tail takes: 3 class slots on each of a group's 16 lanes
constexpr int kStackPostMaxLive = 48;
constexpr int kStackLoopMaxSteps = 128;  // timesteps travel in the kernel arguments
void launch_stack_loop(const FusedLayerSet& ls, int F, int N, int B, int S, int H, int dh, const StackHead& head,
                       const StackLoop& lp, hipStream_t st);
// alignment / overlap scores of decoded layouts (kernels_metrics.hip): bbox [B][S][4] f32, mask [B][S] u8 -> out [B][6];
// -1 if S is outside [1, 256]
int launch_layout_metrics(const float* bbox, const uint8_t* mask, int B, int S, float* out, hipStream_t st);
// ids -> {bbox, label, mask} (kernels_decode.hip); centres: [4][n_bin] f64 cluster centres or nullptr (linear bins)
void launch_decode_layouts(const int32_t* tokens, int B, int E, int A, int n_category, int n_bin,
                           const double* centres, int box_f64, void* bbox, int64_t* label, uint8_t* mask,
                           hipStream_t st);
// cond=relation logit adjustment (kernels_relation.hip)
struct RelArgs {
  float* logp;              // (B,C,S) log p(x_{t-1}|x_t), updated in place
  const int32_t* cond_seq;  // (B,S) conditioned sequence: element e is a graph node iff cond_seq[b][e*A] != pad
  const int32_t* edge_off;  // (B+1) offsets of each layout's edges
  const int32_t *edge_src, *edge_dst, *edge_attr;  // node ids inside the layout's graph (0 = canvas), bitmasks
  const float* centres;     // (4, n_bin) cluster centres, x y w h
  int canvas_bins[4];       // bin of the canvas box (0.5, 0.5, 1, 1) per coordinate
  float step;               // relation_lambda / (14 * number of graphs in the call)
  int num_update, B, C, S, A, n_category, n_bin, pad_id;
  int logp_tm;              // 0: logp is (B,C,S) (the API's layout); 1: (B,S,C), token-major (the handle's own buffer)
};
void launch_relation_update(const RelArgs& a, hipStream_t st);
// the fused tail of an adjusted cond=relation step of the per-step path (kernels_relation.hip relation_step_k): posterior
// (+ strong mask) -> SGD -> [PAD] disable -> draw in ONE launch; p = the step's PostArgs (logits in, tokens out, x_next),
// a = the graph (a.logp unused).  relation_step_supported: S <= 128 and live sub-vocabularies <= 48 classes.
struct PostArgs;
void launch_relation_step(const PostArgs& p, const RelArgs& a, hipStream_t st);
bool relation_step_supported(const PostArgs& p);
// MFMA attention on the head-padded fp16 layout (kernels_attn16.hip)
void launch_attention16(const __half* qkv, __half* out, int B, int S, int H, int dh, int ldq, int ldo, hipStream_t st);

// ---- attention (kernels_attn.hip): softmax(QK^T/sqrt(dh)) V per (layout, head) ------------
struct AttnArgs {
  const void* qkv;  // [M, ld] fp32 or fp16; q cols [0,D), k [D,2D), v [2D,3D); head h = cols h*dh..
  int in_f16;       // dtype of qkv
  float* out32;     // [M, ldo32] or nullptr
  __half* out16;    // [M, ldo16] or nullptr
  __half* out16lo;  // split mode, or nullptr
  int B, S, H, dh, D, ld, ldo32, ldo16;
};
void launch_attention(const AttnArgs& a, hipStream_t st);
// split mode, the reference's head geometry (kernels_attn16.hip): fp32 qkv in, fp16 x 3 MFMAs, hi / lo fp16 out
bool attention16x3_supported(int S, int dh, int D, int ld, int ldo);
void launch_attention16x3(const float* qkv, __half* out_hi, __half* out_lo, int B, int S, int H, int dh, int D, int ld,
                          int ldo, hipStream_t st);

// split mode, r06 (kernels_attnout.hip): attention AND out_proj in one launch, one workgroup per layout:
//   out[row, :] = res[row, :] + bias + out_scale * sum_h softmax(q_h k_h^T * scale) v_h · Wo_h^T
// q / k / v arrive head-padded and PANEL-major from in_proj's epilogue (kernels_lngemm.hip, panel_out): hi and lo fp16 arrays
// [3 * 8 * 2 panels][rows][32], panel (which * 8 + head) * 2 + d / 32, panel_stride bytes apart; the kernel reads up to 128 rows per
// layout (rows S .. 127 belong to the next layout or to the buffer's slack: finite values, masked).
struct AttnOutArgs {
  const char *qkv_hi, *qkv_lo;   // panel arrays
  size_t panel_stride;           // bytes, a multiple of 16
  const char* w_img;             // ldm_pack::pack_x3_kstep_image of out_proj: 32 stages x 32 KiB
  const float* res;              // [M, D] residual rows (AdaLN(x))
  const float* bias;             // [D] out_proj bias
  float* out;                    // [M, D]
  int S, D;
  float scale;                   // 1 / sqrt(head dim)
  float out_scale;               // 2^-k of out_proj's power-of-two pre-scale
  int w2;                        // two-product out_proj (Wo fp16 only); the attention's own products keep all three terms
  // hybrid mode (ffn_img != nullptr; needs w2): the block's plain-fp16 FFN behind the attention in the SAME launch — `out` is not written,
  // ffn_out [M, D] receives x + attention + FFN; ffn_img = ldm_pack::pack_ffn_image_pipelined ((n_chunks + 1) x 64 KiB), F == 32 n_chunks <= 2048
  const char* ffn_img;
  const float *ffn_gamma, *ffn_beta, *ffn_b1, *ffn_b2;
  float* ffn_out;
  int F, n_chunks;
};
bool attnout16x3_supported(int S, int H, int dh, int D);
int launch_attnout16x3(const AttnOutArgs& a, int B, hipStream_t st);   // -1: geometry not supported
void attnout_phase_read(unsigned long long* out24);   // (LDM_ATTNOUT_TM=1: accumulated phase cycles, reset on read)

// ---- posterior + categorical draw (kernels_post.hip) ------------------------------------
struct VocabTables {  // built on the host from the tokenizer geometry (layout_tokenizer.py:429-467)
  int n_class, n_attr, pad_id, mask_id;
  int start[kMaxAttr];  // first full id of the attribute's body
  int count[kMaxAttr];  // body size (n_category or n_bin); K = count + 2
};
struct PostArgs {
  const float* logits;     // [M, ldl] (cols >= n_class ignored)
  int ldl;
  const int32_t* tokens;   // [M] current x_t
  int32_t* tokens_out;     // [M] or nullptr
  float* logp_out;         // (B,C,S) or nullptr (parity hook)
  const float* logp_in;    // (B,C,S) or nullptr: skip the posterior, only sample (ldm_sample_tokens)
  int logp_tm;             // 1: logp_out / logp_in are (B,S,C), token-major (a token's classes contiguous: the handle's own
                           // cond=relation buffer); 0: the API's (B,C,S)
  const float* sched;      // device [8][n_attr][T+1] schedule buffers, see ScheduleRow
  int T;                   // n_step
  int t_post;              // timestep of q_posterior
  const int32_t* cond_seq; // [M] or nullptr
  const uint8_t* strong;   // [M] or nullptr
  const float* weak;       // (B,C,S) or nullptr
  int pad_disable;
  int kind;                // LDM_SAMPLE_*
  float temperature, top_p;
  int top_k;
  const uint64_t* rng;     // device {seed, first_layout}
  int f32_lse;             // 1: fp32 log-softmax (fast numerics mode); 0: float64 like the reference (base.py:137)
  int layout_off;          // + offset of this launch's first layout inside the call's batch
  int step;                // reverse-loop index (RNG counter word)
  int B, S;
  VocabTables v;
  // next step's embedding, fused: x_next[row] = emb[new token] + pos[s] (nn_lib.py:204,220), or nullptr
  float* x_next;
  const float *emb, *pos;
  int D, ldx;
  // near-tie report of deterministic decoding: tie_flags[b] = 1 when some token of layout b was decided with a lead over
  // the runner-up below tie_rel * max |logit of the token| (what the fp16 mode's logits error can move), or nullptr
  // ... or below the absolute floor tie_abs (6 x a bound on the ABSOLUTE logits error of the checkpoint, measured by the
  // caller: the error of a small-magnitude row does not shrink with that row's own max |logit|)
  uint8_t* tie_flags;
  float tie_rel, tie_abs;
};
enum ScheduleRow { kLogAt = 0, kLogBt, kLogCt, kLogCumAt, kLogCumBt, kLogCumCt, kLog1mCt, kLog1mCumCt, kNumSched };
void launch_posterior_sample(const PostArgs& p, hipStream_t st);
void launch_set_rng(uint64_t* rng, uint64_t seed, uint64_t first_layout, hipStream_t st);

// ---- FID feature extractor (kernels_fid.hip): FIDNetV3.extract_features, trainer/fid/model.py:123-164 -------------
struct FidLayer {            // nn.TransformerEncoderLayer(256, 4, 128), weights TRANSPOSED to [K][N]
  const float *in_wt, *in_b;     // [256][768], [768]
  const float *out_wt, *out_b;   // [256][256], [256]
  const float *w1t, *b1;         // [256][128], [128]
  const float *w2t, *b2;         // [128][256], [256]
  const float *n1_g, *n1_b, *n2_g, *n2_b;
};
struct FidArgs {
  const float* bbox;             // (B, N, 4)
  const int64_t* label;          // (B, N)
  const uint8_t* padding_mask;   // (B, N) 1 = padded element
  float* feat;                   // (B, 256)
  const float *emb_label;        // [num_label][256]
  const float *fc_bbox_wt, *fc_bbox_b;  // [4][256], [256]
  const float *fc_in_wt, *fc_in_b;      // [512][256], [256]
  const float* token;            // [256]
  FidLayer layer[8];
  int n_layer, N, num_label;
};
void launch_fid_features(const FidArgs& a, int B, hipStream_t st);

// ---- precision / recall / density / coverage (kernels_prdc.hip): helpers/metric.py:37-59 via prdc ^0.2 ----------------
void launch_prdc_pdist2(const float* A, int n, const float* B, int m, int dim, float* D, hipStream_t st);
void launch_prdc_kth(const float* D, int n, int m, int k1, float* r2, hipStream_t st);
// counts[4] (zeroed by the caller) += {fakes inside a real radius, reals with a fake inside that fake's radius,
// sum over fakes of reals whose radius holds them, reals whose nearest fake is inside their radius}
void launch_prdc_counts(const float* Drf, int n, int m, const float* r2_real, const float* r2_fake, unsigned long long* counts,
                        hipStream_t st);

// ---- small utilities ---------------------------------------------------------------------
void launch_delay_us(int us, hipStream_t st);  // one wave spinning for `us` microseconds (lane phase offset)
// dst = fp16(scale * src) (+ dstlo = the split-mode lo part)
void launch_f32_to_f16(const float* src, __half* dst, __half* dstlo, int64_t n, hipStream_t st, float scale = 1.0f);
// AdaLN table: out[t][l][2D] = Linear(SiLU(Emb[t])) (transformer_utils.py:67-69,80)
void launch_adaln_table(const float* emb /*[T,D]*/, const float* w /*[2D,D]*/, const float* b /*[2D]*/,
                        float* out /*[T, L, 2D] at layer offset*/, int T, int D, int L, int layer, hipStream_t st);
void launch_pos_table(const float* elem /*[E,D]*/, const float* attr /*[A,D]*/, float* pos /*[S,D]*/, int E,
                      int A, int D, hipStream_t st);

}  // namespace ldm
