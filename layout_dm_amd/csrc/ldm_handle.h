// Host side of libldm_hip.so, shared declarations: the handle (packed weights, per-lane activation workspaces, staging
// buffers, hipGraph cache, profiling) and the helpers the translation units of the C-ABI share.
//   ldm_api.cpp      lifecycle, parity hooks, result packaging, near-tie report, introspection
//   ldm_weights.cpp  checkpoint upload, fp16 / LDS weight images, parameter tables (ldm_load_weight / ldm_finalize_weights)
//   ldm_denoise.cpp  the launch sequence of one denoiser pass over a chunk, per numerics mode
//   ldm_loop.cpp     the hot path: one reverse step, the one-launch loop, the per-lane hipGraph loop (ldm_sample_step / _loop)
#pragma once
#include "../../include/ldm_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ldm_kernels.h"
#include "ldm_pack.h"

using namespace ldm;


#define HIP_OK(h, expr)                                                                        \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) return (h)->fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                           __FILE__, __LINE__);                                \
  } while (0)


// Entry points run on the handle's device but leave the calling thread's current device as they found it
// (PyTorch tracks its own notion of the current device per thread).
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
    else if (err == hipSuccess) prev = -1;  // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
// (also drops a stale sticky error of an unrelated earlier runtime call — e.g. the caller's framework probing a host
//  pointer with hipPointerGetAttributes — so that the hipGetLastError() after our launches reports only our own)
#define ON_DEVICE(h)                                                                                    \
  DeviceGuard _dev_guard((h)->device);                                                                  \
  (void)hipGetLastError();                                                                              \
  if (_dev_guard.err != hipSuccess) return (h)->fail(-2, "hipSetDevice(%d) failed: %s", (h)->device,    \
                                                     hipGetErrorString(_dev_guard.err))

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct Raw {  // a checkpoint tensor as uploaded (fp32, device)
  float* d = nullptr;
  std::vector<int64_t> shape;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

struct LayerW {
  const float *w_in, *b_in, *w_out, *b_out, *w1, *b1, *w2, *b2, *g2, *be2;  // fp32 views
  __half *w_in16, *w_in16lo, *w_out16, *w_out16lo, *w1_16, *w1_16lo, *w2_16, *w2_16lo;
  float s_in = 1.f, s_out = 1.f, s1 = 1.f, s2 = 1.f;  // split mode: 2^-k of each tensor's power-of-two pre-scale (GemmArgs.out_scale)
  void *x3_qkv = nullptr, *x3_ffn1 = nullptr;         // split mode: hi | lo tile images of in_proj / linear1 (kernels_lngemm.hip)
  void *x3_out_slab = nullptr, *x3_ffn2_slab = nullptr;   // ... K-slab images of out_proj / linear2 (its GEMM prologue, lngemm level 2)
  // r06, fused attention + out_proj (kernels_attnout.hip): in_proj with head-padded output columns (48 tiles, bias [1536]) and
  // out_proj as a k-step image (ldm_pack::pack_x3_kstep_image)
  void *x3_qkv_pad = nullptr, *x3_out_kstep = nullptr;
  void* ffn16_img = nullptr;   // hybrid: the fused plain-fp16 FFN's chunk image (kernels_ffn16.hip; the fast mode's pack_ffn_image_pipelined)
  float* b_in_pad = nullptr;
};

struct ProfEntry {
  std::string name;
  double ms = 0;
  int64_t launches = 0;
  double flops = 0, bytes = 0;
};

struct PendingEvent {
  int entry;
  hipEvent_t a, b;
};

struct GraphKey {
  int B, n_steps, kind, top_k, has_cond, has_strong, has_weak, pad_disable, has_inter;
  int has_rel = 0, rel_num_update = 0, rel_n_graph = 0, rel_bins[4] = {0, 0, 0, 0};
  float tie_rel = 0.f, tie_abs = 0.f;
  float rel_lambda = 0.f;
  const void* rel_edges = nullptr;
  float temperature, top_p;
  const void *tokens, *cond_seq, *strong, *weak;
  std::vector<int32_t> t_model, t_post;
  bool operator==(const GraphKey& o) const {
    return B == o.B && n_steps == o.n_steps && kind == o.kind && top_k == o.top_k && has_cond == o.has_cond &&
           has_strong == o.has_strong && has_weak == o.has_weak && pad_disable == o.pad_disable &&
           has_inter == o.has_inter && tie_rel == o.tie_rel && tie_abs == o.tie_abs && has_rel == o.has_rel && rel_num_update == o.rel_num_update &&
           rel_n_graph == o.rel_n_graph && rel_lambda == o.rel_lambda && rel_edges == o.rel_edges &&
           rel_bins[0] == o.rel_bins[0] && rel_bins[1] == o.rel_bins[1] && rel_bins[2] == o.rel_bins[2] &&
           rel_bins[3] == o.rel_bins[3] && temperature == o.temperature && top_p == o.top_p && tokens == o.tokens &&
           cond_seq == o.cond_seq && strong == o.strong && weak == o.weak &&
           t_model == o.t_model && t_post == o.t_post;
  }
};

struct GraphEntry {
  GraphKey key;
  std::vector<hipGraph_t> graph;      // one per lane
  std::vector<hipGraphExec_t> exec;
  void destroy() {
    for (auto e : exec)
      if (e) (void)hipGraphExecDestroy(e);
    for (auto g : graph)
      if (g) (void)hipGraphDestroy(g);
    exec.clear();
    graph.clear();
  }
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }


struct ldm_handle {
  ldm_config cfg{};
  int device = 0;
  std::string err;
  // geometry
  int S = 0, C = 0, D = 0, F = 0, H = 0, dh = 0, L = 0, T = 0, Dp = 0, Fp = 0, Cp = 0, chunk = 0;
  VocabTables vocab{};
  // weights
  std::map<std::string, Raw> raw;
  bool finalized = false;
  std::vector<LayerW> layers;
  float *pos = nullptr, *adaln = nullptr, *sched = nullptr;
  const float *emb = nullptr, *head_g = nullptr, *head_b = nullptr, *head_w = nullptr;
  __half *head_w16 = nullptr, *head_w16lo = nullptr;
  float head_s = 1.f;
  // split mode on the reference's backbone: the three LayerNorm-fed GEMMs (AdaLN + in_proj, norm2 + linear1, head LN + head)
  // run as ONE row-resident launch each (kernels_lngemm.hip) instead of a LayerNorm launch + gemm16x3_k
  bool lngemm = false;
  bool lngemm_pre = false;   // level 2: out_proj / linear2 as the GEMM prologue of the row-resident kernel that consumes their sum
  bool pre_out = false, pre_ffn2 = false;   // ... which of the two (dev: LDM_X3_LNGEMM=3 / 4 = only out_proj / only linear2)
  void* x3_head = nullptr;
  int x3_qkv_tiles = 0, x3_ffn1_tiles = 0, x3_head_tiles = 0;
  // r06: attention and out_proj of the split mode as ONE layout-resident launch (kernels_attnout.hip); q / k / v travel from in_proj
  // to it as head-padded hi / lo fp16 PANELS (qkvp_hi / qkvp_lo: [48][panel_rows][32]); LDM_DEV=1 LDM_X3_ATTNOUT=0: attn16x3_k + gemm16x3_k
  bool attnout = false;
  size_t panel_rows = 0;
  // r06: the hidden activations (linear1 -> ReLU -> linear2) travel panel-major as well when linear2 is a GEMM prologue (pre_ffn2):
  // full-line stores in linear1's epilogue, 2-KiB-contiguous A loads in the prologue; LDM_DEV=1 LDM_X3_HIDPANEL=0: row-major
  bool hid_panels = false;
  // two-product form of the split mode's WEIGHT GEMMs: activations hi + lo, weights fp16 only (kernels_lngemm.hip / kernels_attnout.hip W2)
  bool w2p = false;
  // created as LDM_PREC_MIXED_F16 (1) / LDM_PREC_HYBRID_F16 (2): cfg.precision then reads LDM_PREC_SPLIT_F16 — every other choice is the split mode's
  int mixed = 0;
  // products per k16-step (kernels_lngemm.hip NPM / NPP): of the attention path's GEMMs (in_proj; out_proj: w2p) and of linear1 / linear2 / the head.
  // split 3 / 3, mixed 2 / 2, hybrid 2 / 1 (the FFN and the head in plain fp16: LayerNorm output, hidden activations and weights rounded once)
  int np_w = 3, np_ffn = 3;
  // hybrid: linear1 + ReLU + linear2 + residual as ONE plain-fp16 launch per block (kernels_ffn16.hip) instead of linear1 -> plain-fp16 panels ->
  // linear2 as the GEMM prologue of the next launch; LDM_DEV=1 LDM_HYB_FFN=0: the two-launch form
  bool ffn_fused = false;
  // ... and that FFN behind the attention in the SAME launch (kernels_attnout.hip FFN): two launches per block; LDM_DEV=1 LDM_HYB_ATTNFFN=0: three
  bool attn_ffn_fused = false;
  bool balanced_chunks = true;   // ldm_loop.cpp run_loop_body: a call's passes share its layouts evenly; LDM_DEV=1 LDM_BALANCED_CHUNKS=0: full chunks + a remainder
  std::vector<void*> owned;    // everything hipMalloc'ed by the handle for its lifetime
  std::vector<void*> derived;  // what ldm_finalize_weights derives from the checkpoint (fp16 / split copies, LDS images, parameter
                               // tables): freed and rebuilt when the weights are finalized again (a reload used to leak them)
  bool to_derived = false;     // dalloc's destination while ldm_finalize_weights runs
  int n_cu = 256;              // compute units of the device: the batch quantum of the one-launch loop (one workgroup per layout)
  // workspace of ONE chunk.  These are the pointers the launch sequences use; with several lanes (below) they are
  // switched to the lane's own buffers by activate() before its launches are recorded / issued.
  float *P = nullptr, *Q = nullptr, *qkv32 = nullptr, *att32 = nullptr, *h32 = nullptr, *hid32 = nullptr,
        *logits = nullptr;
  __half *a16 = nullptr, *a16lo = nullptr, *qkv16 = nullptr, *att16 = nullptr, *att16lo = nullptr, *h16 = nullptr,
         *h16lo = nullptr, *hid16 = nullptr, *hid16lo = nullptr, *qkvp_hi = nullptr, *qkvp_lo = nullptr;
  // Lanes: chunks c, c + n_lanes, ... form lane (c % n_lanes); every lane has its own workspace, stream and
  // captured graph, and the lanes run CONCURRENTLY, lane l starting l * lane_offset_us late.  Why: the fused
  // kernels alternate HBM-bound phases (row loads / stores, ~30 % of a block) with MFMA-bound phases, and with one
  // kernel on the whole chip every CU hits the memory phase at the same moment (all-CU burst ~4 TB/s, then HBM
  // idles).  Two half-chip kernels out of phase halve each burst (profiles/r02_call2_phase_vs_blocks.txt).
  struct Workspace {
    float *P, *Q, *qkv32, *att32, *h32, *hid32, *logits, *rel_logp;
    __half *a16, *a16lo, *qkv16, *att16, *att16lo, *h16, *h16lo, *hid16, *hid16lo, *qkvp_hi, *qkvp_lo;
    float2 *stats_a, *stats_b;
  };
  std::vector<Workspace> ws;
  std::vector<hipStream_t> lane_stream;
  std::vector<hipEvent_t> lane_done;
  hipEvent_t fork_ev = nullptr;
  int n_lanes = 1, lane_offset_us = 0, cur_lane = -1;
  void save_ws(int l) {
    ws[l] = Workspace{P, Q, qkv32, att32, h32, hid32, logits, rel_logp, a16, a16lo, qkv16, att16, att16lo,
                      h16, h16lo, hid16, hid16lo, qkvp_hi, qkvp_lo, stats_a, stats_b};
  }
  void activate(int l) {
    if (l == cur_lane) return;
    if (cur_lane >= 0) ws[cur_lane].rel_logp = rel_logp;  // (allocated lazily)
    const Workspace& w = ws[l];
    P = w.P; Q = w.Q; qkv32 = w.qkv32; att32 = w.att32; h32 = w.h32; hid32 = w.hid32; logits = w.logits;
    rel_logp = w.rel_logp; a16 = w.a16; a16lo = w.a16lo; qkv16 = w.qkv16; att16 = w.att16; att16lo = w.att16lo;
    h16 = w.h16; h16lo = w.h16lo; hid16 = w.hid16; hid16lo = w.hid16lo; qkvp_hi = w.qkvp_hi; qkvp_lo = w.qkvp_lo;
    stats_a = w.stats_a; stats_b = w.stats_b;
    cur_lane = l;
  }
  // fast-mode (fp16 LDS-DMA GEMM + MFMA attention) layout: K padded to 64, heads padded 58 -> 64
  int Dq = 0, HD = 0, Fq = 0, Mpad = 0;
  int gemm_cfg[5] = {0, 0, 0, 0, 0};  // qkv, attn_out, ffn1, ffn2, head
  struct FastLayer {
    __half *w_in = nullptr, *w_out = nullptr, *w1 = nullptr, *w2 = nullptr;  // head-padded fp16 copies (generic tiled GEMMs)
    void* attn_head_img_ks = nullptr;  // per head: 6 in_proj tiles (k-slot K) + its 2 out-proj slabs (stack kernel)
    void* ffn_img_pipe = nullptr;      // W1 tile i | W2 slab i - 1 per stage: the software-pipelined chunk stream (stack kernel)
    float* b_in = nullptr;
    float* b_out_v = nullptr;  // out_proj bias + W_out b_v (the stack kernel never adds the V bias: softmax rows sum to 1)
  };
  std::vector<FastLayer> fast;
  __half* fast_head = nullptr;
  void* head_img_ks = nullptr;  // vocabulary head as 32-class tile images, K axis in k-slot order (stack kernel)
  float2 *stats_a = nullptr, *stats_b = nullptr;  // deferred normalisation: per-row (mean, rstd) of P / Q
  int fused_attn = 6;  // 6: the layout-resident stack kernel (kernels_stack.hip: all layers + vocabulary head per launch, rows
                       //    in the out-projection accumulators; the reference's backbone on both of its datasets);
                       // 0: generic tiled kernels (LayerNorm -> gemm16 -> attention16 -> ...) for every other accepted
                       //    geometry (and as an A/B / cross-check of the stack kernel: LDM_FUSED_ATTN=0)
  // parameter-table LDS images of the loop kernel (ldm_kernels.h StackTables), built by build_loop_tables
  float *tbl_att_static = nullptr, *tbl_att_dyn = nullptr, *tbl_ffn = nullptr, *tbl_head = nullptr;
  int rel_loop = 1;    // cond=relation inside the one-launch loop (LDM_REL_LOOP=0: the per-step path)
  int stack_loop = 1;  // the WHOLE reverse loop of a layout in its workgroup (kernels_stack.hip HEAD == 2): one launch per
                       // sampling call, the step's tail behind the vocabulary head (LDM_STACK_LOOP=0: one stack launch +
                       // one posterior launch per step, captured in per-lane hipGraphs — the r02 path)
  // near-tie report of deterministic decoding (ldm_set_tie_report): flags [tie_steps][max_batch]
  float tie_rel = 0.f, tie_abs = 0.f;
  uint8_t* tie_flags = nullptr;
  int tie_steps = 0;
  // cond staging (handle-owned, fixed addresses) so a captured graph does not depend on caller pointers
  int32_t* st_cond_seq = nullptr;
  uint8_t* st_strong = nullptr;
  float* st_weak = nullptr;
  int32_t* st_inter = nullptr;  // (n_step, max_batch, S) intermediates of a graph-captured loop
  // cond=relation: the adjusted log-probabilities of one chunk + staging of the caller's graph (fixed addresses)
  float* rel_logp = nullptr;            // (chunk, C, S)
  int32_t* st_rel_off = nullptr;        // (max_batch + 1)
  int32_t* st_rel_edges = nullptr;      // 3 x st_rel_cap : src | dst | attr
  size_t st_rel_cap = 0;
  float* st_rel_centres = nullptr;      // (4, n_bin)
  int32_t *tok_a = nullptr, *tok_b = nullptr;  // loop state ping-pong (max_batch)
  uint64_t* rng = nullptr;                      // device {seed, first_layout}
  // profiling
  bool profiling = false;
  std::vector<ProfEntry> prof;
  std::vector<PendingEvent> pending;
  hipEvent_t loop_a = nullptr, loop_b = nullptr;
  bool loop_timed = false;
  std::vector<GraphEntry> graphs;

  int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }

  template <typename Tp>
  int dalloc(Tp** out, size_t count, bool zero = true) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(Tp), 16);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(-3, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    if (zero) {
      e = hipMemset(p, 0, bytes);
      if (e != hipSuccess) return fail(-3, "hipMemset failed: %s", hipGetErrorString(e));
    }
    (to_derived ? derived : owned).push_back(p);
    *out = reinterpret_cast<Tp*>(p);
    return 0;
  }

  int prof_entry(const char* name) {
    for (size_t i = 0; i < prof.size(); ++i)
      if (prof[i].name == name) return (int)i;
    ProfEntry e;
    e.name = name;
    prof.push_back(e);
    return (int)prof.size() - 1;
  }

  // bracket one launch with events when profiling (never during graph capture)
  struct Scope {
    ldm_handle* h;
    hipStream_t st;
    int entry = -1;
    hipEvent_t a = nullptr, b = nullptr;
    bool ok = false;
    Scope(ldm_handle* h_, hipStream_t st_, const char* name, double flops, double bytes) : h(h_), st(st_) {
      if (!h->profiling) return;
      entry = h->prof_entry(name);
      h->prof[entry].launches += 1;
      h->prof[entry].flops += flops;
      h->prof[entry].bytes += bytes;
      // a failed event call only loses this timing sample (ok stays false); the launch itself is unaffected
      ok = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess && hipEventRecord(a, st) == hipSuccess;
    }
    ~Scope() {
      if (entry < 0) return;
      if (ok && hipEventRecord(b, st) == hipSuccess) {
        h->pending.push_back({entry, a, b});
        return;
      }
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  };

  void drain_profile() {
    for (auto& pe : pending) {
      float ms = 0;
      if (hipEventSynchronize(pe.b) == hipSuccess && hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess)
        prof[pe.entry].ms += ms;
      (void)hipEventDestroy(pe.a);
      (void)hipEventDestroy(pe.b);
    }
    pending.clear();
  }
};

// ---- shared between the translation units (definitions: see the list at the top)
namespace ldm_host {
std::string& create_error();   // last ldm_create error of this thread (ldm_last_error(NULL))
double gemm_flops(int M, int N, int K);
int denoise_chunk(ldm_handle* h, const int32_t* d_tokens, int t, int Bc, hipStream_t st, bool skip_embed = false);
void fill_post(ldm_handle* h, ldm::PostArgs& p, const ldm_cond* cond, const ldm_sampler* s, size_t layout_off, int Bc);
int check_ready(ldm_handle* h, int B);
int check_sampler(ldm_handle* h, const ldm_sampler* s);
int set_rng(ldm_handle* h, uint64_t seed, uint64_t first_layout, hipStream_t st);
void fill_rel(ldm_handle* h, ldm::RelArgs& a, const ldm_relation* rel, size_t layout_off, int Bc);
bool loop_fusable(const ldm_handle* h, const ldm_relation* rel);
}  // namespace ldm_host

