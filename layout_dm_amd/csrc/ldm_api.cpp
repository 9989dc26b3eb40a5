// Host side of libldm_hip.so: the C-ABI declared in include/ldm_hip.h.
// Owns the repacked weights, the per-chunk activation workspace (sized so the working set of one
// pass through the network stays inside the 256 MiB Infinity Cache), the launch sequence of one
// reverse step and the hipGraph cache for the T-step loop.  No torch types cross this boundary.
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

static thread_local std::string g_create_error;

#define HIP_OK(h, expr)                                                                        \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) return (h)->fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                           __FILE__, __LINE__);                                \
  } while (0)

namespace {

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

}  // namespace

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
  std::vector<void*> owned;  // everything hipMalloc'ed by the handle
  // workspace of ONE chunk.  These are the pointers the launch sequences use; with several lanes (below) they are
  // switched to the lane's own buffers by activate() before its launches are recorded / issued.
  float *P = nullptr, *Q = nullptr, *qkv32 = nullptr, *att32 = nullptr, *h32 = nullptr, *hid32 = nullptr,
        *logits = nullptr;
  __half *a16 = nullptr, *a16lo = nullptr, *qkv16 = nullptr, *att16 = nullptr, *att16lo = nullptr, *h16 = nullptr,
         *h16lo = nullptr, *hid16 = nullptr, *hid16lo = nullptr;
  // Lanes: chunks c, c + n_lanes, ... form lane (c % n_lanes); every lane has its own workspace, stream and
  // captured graph, and the lanes run CONCURRENTLY, lane l starting l * lane_offset_us late.  Why: the fused
  // kernels alternate HBM-bound phases (row loads / stores, ~30 % of a block) with MFMA-bound phases, and with one
  // kernel on the whole chip every CU hits the memory phase at the same moment (all-CU burst ~4 TB/s, then HBM
  // idles).  Two half-chip kernels out of phase halve each burst (profiles/r02_call2_phase_vs_blocks.txt).
  struct Workspace {
    float *P, *Q, *qkv32, *att32, *h32, *hid32, *logits, *rel_logp;
    __half *a16, *a16lo, *qkv16, *att16, *att16lo, *h16, *h16lo, *hid16, *hid16lo;
    float2 *stats_a, *stats_b;
  };
  std::vector<Workspace> ws;
  std::vector<hipStream_t> lane_stream;
  std::vector<hipEvent_t> lane_done;
  hipEvent_t fork_ev = nullptr;
  int n_lanes = 1, lane_offset_us = 0, cur_lane = -1;
  void save_ws(int l) {
    ws[l] = Workspace{P, Q, qkv32, att32, h32, hid32, logits, rel_logp, a16, a16lo, qkv16, att16, att16lo,
                      h16, h16lo, hid16, hid16lo, stats_a, stats_b};
  }
  void activate(int l) {
    if (l == cur_lane) return;
    if (cur_lane >= 0) ws[cur_lane].rel_logp = rel_logp;  // (allocated lazily)
    const Workspace& w = ws[l];
    P = w.P; Q = w.Q; qkv32 = w.qkv32; att32 = w.att32; h32 = w.h32; hid32 = w.hid32; logits = w.logits;
    rel_logp = w.rel_logp; a16 = w.a16; a16lo = w.a16lo; qkv16 = w.qkv16; att16 = w.att16; att16lo = w.att16lo;
    h16 = w.h16; h16lo = w.h16lo; hid16 = w.hid16; hid16lo = w.hid16lo; stats_a = w.stats_a; stats_b = w.stats_b;
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
    owned.push_back(p);
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

// ------------------------------------------------------------------------------------------ create
extern "C" int ldm_abi_version(void) { return LDM_ABI_VERSION; }

extern "C" int ldm_get_layout(const ldm_handle* h, int* chunk, int* lanes) {
  if (!h) return -1;
  if (chunk) *chunk = h->chunk;
  if (lanes) *lanes = h->n_lanes;
  return 0;
}

extern "C" const char* ldm_last_error(const ldm_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int ldm_create(const ldm_config* cfg, int device, ldm_handle** out) {
  auto bad = [&](const char* msg) {
    g_create_error = msg;
    return -1;
  };
  if (!cfg || !out) return bad("null argument");
  if (cfg->abi_version != LDM_ABI_VERSION) return bad("ldm_config.abi_version mismatch");
  if (cfg->n_attr != 5) return bad("only the c-x-y-w-h (5 attribute) vocabulary is supported");
  if (cfg->n_category < 1 || cfg->n_bin < 1 || cfg->n_layer < 1 || cfg->n_step < 1 || cfg->n_head < 1 || cfg->d_model < 16)
    return bad("n_category / n_bin / n_layer / n_step / n_head must be >= 1, d_model >= 16");
  if (cfg->n_category + 4 * cfg->n_bin + 2 > 192)
    return bad("vocabulary > 192 classes not supported by the fused posterior kernel");
  if (cfg->d_model % 16 || cfg->d_ff % 16 || cfg->d_model > 1024) return bad("d_model/d_ff must be multiples of 16, d_model <= 1024");
  if (cfg->d_model % cfg->n_head) return bad("d_model must be divisible by n_head");
  if (cfg->d_model / cfg->n_head > 64) return bad("head_dim > 64 not supported");
  if (cfg->precision < 0 || cfg->precision > 2) return bad("unknown precision mode");
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
  h->S = cfg->max_elem * cfg->n_attr;
  h->C = cfg->n_category + 4 * cfg->n_bin + 2;
  h->D = cfg->d_model;
  h->F = cfg->d_ff;
  h->H = cfg->n_head;
  h->dh = h->D / h->H;
  h->L = cfg->n_layer;
  h->T = cfg->n_step;
  h->Dp = round_up(h->D, 32);
  h->Fp = round_up(h->F, 32);
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
    A(&h->a16, Mc * h->Dp);
    A(&h->att16, Mc * h->Dp);
    A(&h->h16, Mc * h->Dp);
    A(&h->hid16, Mc * h->Fp);
    {
      A(&h->qkv32, Mc * 3 * h->D);
      A(&h->a16lo, Mc * h->Dp);
      A(&h->att16lo, Mc * h->Dp);
      A(&h->h16lo, Mc * h->Dp);
      A(&h->hid16lo, Mc * h->Fp);
    }
  }
  h->save_ws(lane);
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
    g_create_error = h->err;
    ldm_destroy(h);
    return rc;
  }
  if (hipEventCreate(&h->loop_a) != hipSuccess || hipEventCreate(&h->loop_b) != hipSuccess ||
      hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming) != hipSuccess) {
    g_create_error = "event creation failed";
    ldm_destroy(h);
    return -2;
  }
  h->lane_stream.assign(h->n_lanes, nullptr);
  h->lane_done.assign(h->n_lanes, nullptr);
  for (int l = 1; l < h->n_lanes; ++l) {
    if (hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->lane_done[l], hipEventDisableTiming) != hipSuccess) {
      g_create_error = "lane stream / event creation failed";
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
  if (h->loop_a) (void)hipEventDestroy(h->loop_a);
  if (h->loop_b) (void)hipEventDestroy(h->loop_b);
  if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
  for (auto st : h->lane_stream)
    if (st) (void)hipStreamDestroy(st);
  for (auto ev : h->lane_done)
    if (ev) (void)hipEventDestroy(ev);
  delete h;
}

// ------------------------------------------------------------------------------------------ weights
static std::string strip_prefix(const char* key) {
  std::string k(key);
  for (const char* p : {"model.module.", "module.", "model."}) {
    const size_t n = strlen(p);
    if (k.compare(0, n, p) == 0) {
      k = k.substr(n);
      break;
    }
  }
  return k;
}

extern "C" int ldm_load_weight(ldm_handle* h, const char* key, const float* h_data, const int64_t* shape, int ndim) {
  if (!h || !key || !h_data || (ndim > 0 && !shape)) return h ? h->fail(-1, "null argument") : -1;
  ON_DEVICE(h);
  const std::string k = strip_prefix(key);
  Raw r;
  r.shape.assign(shape, shape + ndim);
  const int64_t n = r.numel();
  if (n <= 0) return h->fail(-1, "empty tensor for key %s", key);
  auto it = h->raw.find(k);
  if (it != h->raw.end()) {
    if (it->second.numel() != n) return h->fail(-1, "key %s reloaded with a different size", key);
    r.d = it->second.d;
  } else {
    HIP_OK(h, hipMalloc((void**)&r.d, n * sizeof(float)));
  }
  HIP_OK(h, hipMemcpy(r.d, h_data, n * sizeof(float), hipMemcpyHostToDevice));
  h->raw[k] = r;
  h->finalized = false;
  return 0;
}

static int need(ldm_handle* h, const std::string& key, std::initializer_list<int64_t> shape, const float** out) {
  auto it = h->raw.find(key);
  if (it == h->raw.end()) return h->fail(-4, "missing checkpoint key: %s", key.c_str());
  std::vector<int64_t> want(shape);
  if (it->second.shape != want) {
    std::string got;
    for (auto s : it->second.shape) got += std::to_string(s) + ",";
    return h->fail(-4, "checkpoint key %s has shape (%s) — does not match the configured geometry", key.c_str(),
                   got.c_str());
  }
  *out = it->second.d;
  return 0;
}

// fp16 (and split-lo) copy of a [N,K] weight with the K axis zero-padded to Kp
static int make_w16(ldm_handle* h, const float* w, int N, int K, int Kp, __half** hi, __half** lo) {
  const bool split = h->cfg.precision == LDM_PREC_SPLIT_F16;
  int rc = h->dalloc(hi, (size_t)N * Kp);
  if (rc) return rc;
  if (split && (rc = h->dalloc(lo, (size_t)N * Kp))) return rc;
  if (K == Kp) {
    launch_f32_to_f16(w, *hi, split ? *lo : nullptr, (int64_t)N * K, 0);
  } else {
    __half *thi = nullptr, *tlo = nullptr;
    if ((rc = h->dalloc(&thi, (size_t)N * K))) return rc;
    if (split && (rc = h->dalloc(&tlo, (size_t)N * K))) return rc;
    launch_f32_to_f16(w, thi, tlo, (int64_t)N * K, 0);
    HIP_OK(h, hipMemcpy2DAsync(*hi, (size_t)Kp * 2, thi, (size_t)K * 2, (size_t)K * 2, N, hipMemcpyDeviceToDevice, 0));
    if (split)
      HIP_OK(h, hipMemcpy2DAsync(*lo, (size_t)Kp * 2, tlo, (size_t)K * 2, (size_t)K * 2, N, hipMemcpyDeviceToDevice, 0));
  }
  return 0;
}

// ---- fast-mode weight images (built on the host once; tiny compared with one sampling call)
static uint16_t f2h_bits(float x) {
  const __half hh = __float2half(x);
  uint16_t u;
  memcpy(&u, &hh, 2);
  return u;
}

// dst[Np][Kp] fp16 (zero filled) with dst[rmap(n)][cmap(k)] = src[n][k]
template <typename RM, typename CM>
static int pack_w16(ldm_handle* h, const float* d_src, int N, int K, int Np, int Kp, RM rmap, CM cmap, __half** out) {
  std::vector<float> src((size_t)N * K);
  HIP_OK(h, hipMemcpy(src.data(), d_src, src.size() * sizeof(float), hipMemcpyDeviceToHost));
  std::vector<uint16_t> dst((size_t)Np * Kp, 0);
  for (int n = 0; n < N; ++n) {
    const size_t ro = (size_t)rmap(n) * Kp;
    for (int k = 0; k < K; ++k) dst[ro + cmap(k)] = f2h_bits(src[(size_t)n * K + k]);
  }
  int rc = h->dalloc(out, dst.size(), false);
  if (rc) return rc;
  HIP_OK(h, hipMemcpy(*out, dst.data(), dst.size() * 2, hipMemcpyHostToDevice));
  return 0;
}

// ---- LDS-image weight streams: the stack kernel
// copies their weights global -> LDS with linear 1-KiB DMA instructions, so the global copy is stored in
// consumption order with the LDS bank swizzle already applied.
static std::vector<uint16_t> download16(ldm_handle* h, const __half* d, size_t n, int* rc) {
  std::vector<uint16_t> v(n);
  *rc = 0;
  if (hipMemcpy(v.data(), d, n * 2, hipMemcpyDeviceToHost) != hipSuccess) {
    h->err = "hipMemcpy (weight image) failed";
    *rc = -2;
  }
  return v;
}
static int upload_image(ldm_handle* h, const std::vector<uint16_t>& img, void** out) {
  __half* d = nullptr;
  int rc = h->dalloc(&d, img.size(), false);
  if (rc) return rc;
  HIP_OK(h, hipMemcpy(d, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  *out = d;
  return 0;
}
// (index maps and image packers: ldm_pack.h — pure C++, unit-tested on the CPU by tests/cpu_pack_check.cpp)

static int build_fast_weights(ldm_handle* h) {
  const int D = h->D, F = h->F, C = h->C, H = h->H, dh = h->dh, HD = h->HD, Dq = h->Dq, Fq = h->Fq;
  auto id = [](int x) { return x; };
  // in_proj row n = which*D + head*dh + d  ->  (which*H + head)*64 + d   (head slices padded to 64)
  auto qkv_row = [=](int n) { return ldm_pack::qkv_row(n, D, H, dh); };
  // out_proj column k = head*dh + d -> head*64 + d (matches the attention kernel's output layout)
  auto head_col = [=](int k) { return ldm_pack::head_col(k, dh); };
  auto kslot = [](int k) { return ldm_pack::kslot(k); };
  const bool stack = h->fused_attn == 6;  // (geometry checked in ldm_create)
  h->fast.assign(h->L, ldm_handle::FastLayer{});
  int rc;
  for (int i = 0; i < h->L; ++i) {
    const LayerW& w = h->layers[i];
    ldm_handle::FastLayer& f = h->fast[i];
    if (!stack) {  // head-padded fp16 copies for the generic tiled GEMMs + attention16
      if ((rc = pack_w16(h, w.w_in, 3 * D, D, round_up(3 * HD, 256), Dq, qkv_row, id, &f.w_in))) return rc;
      if ((rc = pack_w16(h, w.w_out, D, D, round_up(D, 256), HD, id, head_col, &f.w_out))) return rc;
      if ((rc = pack_w16(h, w.w1, F, D, round_up(F, 256), Dq, id, id, &f.w1))) return rc;
      if ((rc = pack_w16(h, w.w2, D, F, round_up(D, 256), Fq, id, id, &f.w2))) return rc;
    } else {
      // LDS images of the stack kernel.  K axes in MFMA k-slot order (position 16s + 8g + e <- index 16s + 8(e>>2) + 4g +
      // (e&3)): a lane's accumulator-layout registers of column groups 2ks, 2ks+1 ARE its fragment of k16-step ks
      __half *w1p = nullptr, *w2p = nullptr, *w_in_ks = nullptr, *w_out_ks = nullptr;
      auto head_kslot = [=](int k) { return kslot(head_col(k)); };
      if ((rc = pack_w16(h, w.w1, F, D, round_up(F, 256), Dq, id, kslot, &w1p))) return rc;
      if ((rc = pack_w16(h, w.w2, D, F, round_up(D, 256), Fq, id, kslot, &w2p))) return rc;
      if ((rc = pack_w16(h, w.w_in, 3 * D, D, round_up(3 * HD, 256), Dq, qkv_row, kslot, &w_in_ks))) return rc;
      if ((rc = pack_w16(h, w.w_out, D, D, round_up(D, 256), HD, id, head_kslot, &w_out_ks))) return rc;
      const std::vector<uint16_t> h1p = download16(h, w1p, (size_t)F * Dq, &rc);
      if (rc) return rc;
      const std::vector<uint16_t> h2 = download16(h, w2p, (size_t)round_up(D, 256) * Fq, &rc);
      if (rc) return rc;
      const std::vector<uint16_t> ffn = ldm_pack::pack_ffn_image(h1p.data(), h2.data(), Fq, F, 480);
      if ((rc = upload_image(h, ldm_pack::pack_ffn_image_pipelined(ffn, F / 32), &f.ffn_img_pipe))) return rc;
      const std::vector<uint16_t> hin_ks = download16(h, w_in_ks, (size_t)3 * HD * Dq, &rc);
      if (rc) return rc;
      const std::vector<uint16_t> hout = download16(h, w_out_ks, (size_t)round_up(D, 256) * HD, &rc);
      if (rc) return rc;
      const std::vector<uint16_t> slab_ks = ldm_pack::pack_attn_slab_image(hin_ks.data(), hout.data(), H);
      if ((rc = upload_image(h, ldm_pack::pack_attn_head_image(slab_ks, H), &f.attn_head_img_ks))) return rc;
    }
    std::vector<float> b(3 * D), bp((size_t)3 * HD, 0.f);
    HIP_OK(h, hipMemcpy(b.data(), w.b_in, b.size() * 4, hipMemcpyDeviceToHost));
    for (int n = 0; n < 3 * D; ++n) bp[qkv_row(n)] = b[n];
    if ((rc = h->dalloc(&f.b_in, bp.size(), false))) return rc;
    HIP_OK(h, hipMemcpy(f.b_in, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    if (stack) {
      // softmax rows sum to 1, so P (V + 1 b_v^T) = P V + 1 b_v^T and the V bias reaches the block output as the
      // constant W_out b_v: folded into the out-projection bias once, here (fp64 accumulate)
      std::vector<float> wo((size_t)D * D), bo(D), bov(D);
      HIP_OK(h, hipMemcpy(wo.data(), w.w_out, wo.size() * 4, hipMemcpyDeviceToHost));
      HIP_OK(h, hipMemcpy(bo.data(), w.b_out, bo.size() * 4, hipMemcpyDeviceToHost));
      for (int n = 0; n < D; ++n) {
        double acc = bo[n];
        for (int k = 0; k < D; ++k) acc += (double)wo[(size_t)n * D + k] * (double)b[2 * D + k];
        bov[n] = (float)acc;
      }
      if ((rc = h->dalloc(&f.b_out_v, bov.size(), false))) return rc;
      HIP_OK(h, hipMemcpy(f.b_out_v, bov.data(), bov.size() * 4, hipMemcpyHostToDevice));
    }
  }
  if (stack) {
    __half* hk = nullptr;
    if ((rc = pack_w16(h, h->head_w, C, D, round_up(C, 256), Dq, id, kslot, &hk))) return rc;
    const std::vector<uint16_t> hh = download16(h, hk, (size_t)round_up(C, 256) * Dq, &rc);
    if (rc) return rc;
    return upload_image(h, ldm_pack::pack_head_image(hh.data(), h->Cp / 32), &h->head_img_ks);
  }
  return pack_w16(h, h->head_w, C, D, round_up(C, 256), Dq, id, id, &h->fast_head);
}

// Parameter tables of the loop kernel as LDS images (kernels_stack.hip HEAD == 2 copies them global -> LDS with the DMA,
// one phase ahead of their use, instead of 26 loads per thread behind a barrier at every layer entry):
//   att_static[l]      kStackTblAttStatic floats   head-padded in_proj bias
//   att_dyn[t][l]      kStackTblAttDyn floats      1 + AdaLN scale | AdaLN shift | b_out + W_out b_v + shift   (0 beyond d_model)
//   ffn[l]             kStackTblFfn floats         linear1 bias (0-padded to 2048) | norm2 gamma | norm2 beta | linear2 bias
//   head               kStackTblAttDyn floats      head LayerNorm gamma | beta | 0      (takes the att_dyn slot behind the last layer)
static int build_loop_tables(ldm_handle* h) {
  const int D = h->D, F = h->F, L = h->L, T = h->T;
  if (h->fused_attn != 6 || h->H * 64 * 3 != kStackTblAttStatic || D > 512 || F > 2048) return 0;  // not on the stack kernel
  std::vector<float> ada((size_t)T * L * 2 * D);
  HIP_OK(h, hipDeviceSynchronize());  // (the AdaLN table kernels)
  HIP_OK(h, hipMemcpy(ada.data(), h->adaln, ada.size() * 4, hipMemcpyDeviceToHost));
  auto pull = [&](const float* d, size_t n, std::vector<float>& out) -> int {
    out.resize(n);
    HIP_OK(h, hipMemcpy(out.data(), d, n * 4, hipMemcpyDeviceToHost));
    return 0;
  };
  std::vector<float> att_static((size_t)L * kStackTblAttStatic, 0.f), att_dyn((size_t)T * L * kStackTblAttDyn, 0.f),
      ffn((size_t)L * kStackTblFfn, 0.f), head(kStackTblAttDyn, 0.f), v;
  int rc;
  for (int l = 0; l < L; ++l) {
    if ((rc = pull(h->fast[l].b_in, kStackTblAttStatic, v))) return rc;
    std::copy(v.begin(), v.end(), att_static.begin() + (size_t)l * kStackTblAttStatic);
    std::vector<float> bov;
    if ((rc = pull(h->fast[l].b_out_v, D, bov))) return rc;
    for (int t = 0; t < T; ++t) {
      const float* ss = &ada[((size_t)t * L + l) * 2 * D];
      float* o = &att_dyn[((size_t)t * L + l) * kStackTblAttDyn];
      for (int i = 0; i < D; ++i) {
        o[i] = 1.0f + ss[i];           // multiplier (0 beyond d_model: padded columns come out as exact zeros)
        o[512 + i] = ss[D + i];        // shift
        o[1024 + i] = bov[i] + ss[D + i];
      }
    }
    float* f = &ffn[(size_t)l * kStackTblFfn];
    if ((rc = pull(h->layers[l].b1, F, v))) return rc;
    std::copy(v.begin(), v.end(), f);
    if ((rc = pull(h->layers[l].g2, D, v))) return rc;
    std::copy(v.begin(), v.end(), f + 2048);
    if ((rc = pull(h->layers[l].be2, D, v))) return rc;
    std::copy(v.begin(), v.end(), f + 2048 + 512);
    if ((rc = pull(h->layers[l].b2, D, v))) return rc;
    std::copy(v.begin(), v.end(), f + 2048 + 1024);
  }
  if ((rc = pull(h->head_g, D, v))) return rc;
  std::copy(v.begin(), v.end(), head.begin());
  if ((rc = pull(h->head_b, D, v))) return rc;
  std::copy(v.begin(), v.end(), head.begin() + 512);
  auto push = [&](const std::vector<float>& src, float** dst) -> int {
    if (!*dst && (rc = h->dalloc(dst, src.size(), false))) return rc;
    HIP_OK(h, hipMemcpy(*dst, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    return 0;
  };
  if ((rc = push(att_static, &h->tbl_att_static))) return rc;
  if ((rc = push(att_dyn, &h->tbl_att_dyn))) return rc;
  if ((rc = push(ffn, &h->tbl_ffn))) return rc;
  return push(head, &h->tbl_head);
}

extern "C" int ldm_finalize_weights(ldm_handle* h) {
  if (!h) return -1;
  ON_DEVICE(h);
  const int D = h->D, F = h->F, C = h->C, T = h->T, L = h->L;
  const std::string tr = "transformer.";
  int rc;
  const float *elem = nullptr, *attr = nullptr;
  if ((rc = need(h, tr + "cat_emb.weight", {C, D}, &h->emb))) return rc;
  if ((rc = need(h, tr + "pos_emb.elem_emb", {h->cfg.max_elem, D}, &elem))) return rc;
  if ((rc = need(h, tr + "pos_emb.attr_emb", {h->cfg.n_attr, D}, &attr))) return rc;
  if ((rc = need(h, tr + "head.0.weight", {D}, &h->head_g))) return rc;
  if ((rc = need(h, tr + "head.0.bias", {D}, &h->head_b))) return rc;
  if ((rc = need(h, tr + "head.1.weight", {C, D}, &h->head_w))) return rc;
  if (!h->pos && (rc = h->dalloc(&h->pos, (size_t)h->S * D))) return rc;
  if (!h->adaln && (rc = h->dalloc(&h->adaln, (size_t)T * L * 2 * D))) return rc;
  launch_pos_table(elem, attr, h->pos, h->cfg.max_elem, h->cfg.n_attr, D, 0);
  const bool f16 = h->cfg.precision != LDM_PREC_EXACT_F32;
  h->layers.assign(L, LayerW{});
  for (int i = 0; i < L; ++i) {
    const std::string b = tr + "backbone.layers." + std::to_string(i) + ".";
    LayerW& w = h->layers[i];
    const float *emb_t = nullptr, *lin_w = nullptr, *lin_b = nullptr;
    if ((rc = need(h, b + "self_attn.in_proj_weight", {3 * D, D}, &w.w_in))) return rc;
    if ((rc = need(h, b + "self_attn.in_proj_bias", {3 * D}, &w.b_in))) return rc;
    if ((rc = need(h, b + "self_attn.out_proj.weight", {D, D}, &w.w_out))) return rc;
    if ((rc = need(h, b + "self_attn.out_proj.bias", {D}, &w.b_out))) return rc;
    if ((rc = need(h, b + "linear1.weight", {F, D}, &w.w1))) return rc;
    if ((rc = need(h, b + "linear1.bias", {F}, &w.b1))) return rc;
    if ((rc = need(h, b + "linear2.weight", {D, F}, &w.w2))) return rc;
    if ((rc = need(h, b + "linear2.bias", {D}, &w.b2))) return rc;
    if ((rc = need(h, b + "norm1.emb.weight", {T, D}, &emb_t))) return rc;
    if ((rc = need(h, b + "norm1.linear.weight", {2 * D, D}, &lin_w))) return rc;
    if ((rc = need(h, b + "norm1.linear.bias", {2 * D}, &lin_b))) return rc;
    if ((rc = need(h, b + "norm2.weight", {D}, &w.g2))) return rc;
    if ((rc = need(h, b + "norm2.bias", {D}, &w.be2))) return rc;
    launch_adaln_table(emb_t, lin_w, lin_b, h->adaln, T, D, L, i, 0);
    if (f16 && h->cfg.precision != LDM_PREC_FAST_F16) {
      if ((rc = make_w16(h, w.w_in, 3 * D, D, h->Dp, &w.w_in16, &w.w_in16lo))) return rc;
      if ((rc = make_w16(h, w.w_out, D, D, h->Dp, &w.w_out16, &w.w_out16lo))) return rc;
      if ((rc = make_w16(h, w.w1, F, D, h->Dp, &w.w1_16, &w.w1_16lo))) return rc;
      if ((rc = make_w16(h, w.w2, D, F, h->Fp, &w.w2_16, &w.w2_16lo))) return rc;
    }
  }
  if (h->cfg.precision == LDM_PREC_FAST_F16) {
    if ((rc = build_fast_weights(h))) return rc;
    if ((rc = build_loop_tables(h))) return rc;
  } else if (f16 && (rc = make_w16(h, h->head_w, C, D, h->Dp, &h->head_w16, &h->head_w16lo))) {
    return rc;
  }
  // schedule buffers are taken from the checkpoint, not recomputed
  static const char* names[kNumSched] = {"log_at",         "log_bt",         "log_ct",       "log_cumprod_at",
                                         "log_cumprod_bt", "log_cumprod_ct", "log_1_min_ct", "log_1_min_cumprod_ct"};
  static const char* keys[5] = {"c", "x", "y", "w", "h"};
  std::vector<float> host((size_t)kNumSched * h->cfg.n_attr * (T + 1), 0.f);
  for (int k = 0; k < kNumSched; ++k) {
    const bool cum = (k == kLogCumAt || k == kLogCumBt || k == kLogCumCt || k == kLog1mCumCt);
    for (int a = 0; a < h->cfg.n_attr; ++a) {
      // vanilla.py:66-73 registers ONE un-prefixed set; it is replicated into every attribute's row
      const std::string key = h->cfg.q_type == LDM_Q_VANILLA ? std::string(names[k]) : std::string(keys[a]) + "_" + names[k];
      const float* d = nullptr;
      if ((rc = need(h, key, {cum ? T + 1 : T}, &d))) return rc;
      HIP_OK(h, hipMemcpy(&host[((size_t)k * h->cfg.n_attr + a) * (T + 1)], d, (cum ? T + 1 : T) * sizeof(float),
                          hipMemcpyDeviceToHost));
    }
  }
  HIP_OK(h, hipMemcpy(h->sched, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_OK(h, hipDeviceSynchronize());
  HIP_OK(h, hipGetLastError());
  // graphs captured against older weights stay valid (pointers unchanged) but drop them anyway
  for (auto& g : h->graphs) g.destroy();
  h->graphs.clear();
  h->finalized = true;
  return 0;
}

// ------------------------------------------------------------------------------------------ one pass
static double gemm_flops(int M, int N, int K) { return 2.0 * M * N * K; }

// fast mode on the reference's backbone: the stack kernel (kernels_stack.hip) — ONE launch for all layers and the
// vocabulary head, a layout's rows in its workgroup's out-projection accumulators from the embedding output to the
// logits.  Normalisation is deferred into the kernel (no LayerNorm launch, no LN output tensor): the embedding writes raw
// rows, the kernel computes its own row statistics.  (The one-launch reverse loop, run_loop_fused, does not come here: it
// gathers the embedding itself.)
static int denoise_chunk_stack(ldm_handle* h, const int32_t* d_tokens, int t, int Bc, hipStream_t st, bool skip_embed) {
  const int M = Bc * h->S, D = h->D, F = h->F;
  if (!skip_embed) {  // x0 = emb[token] + pos -> P (raw)   (skipped when the previous step's posterior wrote P)
    LnArgs a{};
    a.tokens = d_tokens; a.emb = h->emb; a.pos = h->pos; a.y32 = h->P; a.stats_out = h->stats_a; a.raw = 1;
    a.M = M; a.D = D; a.S = h->S; a.ld16 = h->Dq;
    ldm_handle::Scope sc(h, st, "embed_stats", 0, (double)M * D * 8);
    launch_layernorm(a, st);
  }
  FusedLayerSet ls{};
  ls.n_layer = h->L;
  for (int i = 0; i < h->L; ++i) {
    const LayerW& w = h->layers[i];
    const float* ss = h->adaln + ((size_t)t * h->L + i) * 2 * D;
    ls.w[i] = FusedLayerW{h->fast[i].attn_head_img_ks, h->fast[i].b_in, ss, ss + D, h->fast[i].b_out_v,
                          h->fast[i].ffn_img_pipe, w.b1, w.b2, w.g2, w.be2};
  }
  const StackHead hd{h->head_img_ks, h->head_g, h->head_b, h->logits, h->Cp, h->Cp / 32};
  ldm_handle::Scope sc(h, st, "layers_fused",
                       h->L * (gemm_flops(M, 3 * D, D) + 4.0 * Bc * h->H * (double)h->S * h->S * h->dh +
                               gemm_flops(M, D, D) + 2 * gemm_flops(M, F, D)) + gemm_flops(M, h->C, D),
                       (double)M * (D * 4 + h->Cp * 4));
  launch_stack_stream(ls, F, h->P, D, Bc, h->S, h->H, h->dh, hd, st);
  return 0;
}

// fast mode, every other accepted geometry: fp16 LDS-DMA GEMMs + MFMA attention on the head-padded layout
static int denoise_chunk_fast(ldm_handle* h, const int32_t* d_tokens, int t, int Bc, hipStream_t st,
                              bool skip_embed = false) {
  if (h->fused_attn == 6) return denoise_chunk_stack(h, d_tokens, t, Bc, st, skip_embed);
  const int M = Bc * h->S, D = h->D, F = h->F, C = h->C, Dq = h->Dq, HD = h->HD, Fq = h->Fq;
  auto gemm = [&](const char* name, int tag, const __half* A, int lda, int K, const __half* W, int ldw, int N,
                  const float* bias, int relu, const float* res, float* C32, int ldc32, __half* C16, int ldc16,
                  double flops, double bytes) {
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.relu = relu; g.res = res; g.ldres = D;
    g.C32 = C32; g.ldc32 = ldc32; g.C16 = C16; g.ldc16 = ldc16;
    const int cfg = h->gemm_cfg[tag];
    g.M = M; g.N = N; g.K = round_up(K, gemm16_block_k(cfg)); g.lda = lda; g.ldw = ldw; g.precision = 1;
    ldm_handle::Scope sc(h, st, name, flops, bytes);
    launch_gemm16(g, cfg, tag, st);
  };
  for (int i = 0; i < h->L; ++i) {
    const LayerW& w = h->layers[i];
    const ldm_handle::FastLayer& f = h->fast[i];
    const float* ss = h->adaln + ((size_t)t * h->L + i) * 2 * D;
    {
      LnArgs a{};
      a.x = h->P; a.tokens = (i == 0) ? d_tokens : nullptr; a.emb = h->emb; a.pos = h->pos;
      a.p0 = ss; a.p1 = ss + D; a.y32 = h->P; a.y16 = h->a16;
      a.M = M; a.D = D; a.S = h->S; a.ld16 = Dq; a.ada = 1;
      ldm_handle::Scope sc(h, st, i == 0 ? "embed_adaln" : "adaln", 0, (double)M * D * 10);
      launch_layernorm(a, st);
    }
    gemm("gemm_qkv", 0, h->a16, Dq, D, f.w_in, Dq, 3 * HD, f.b_in, 0, nullptr, nullptr, 0, h->qkv16,
         3 * HD, gemm_flops(M, 3 * D, D), (double)M * (D * 2 + 3 * HD * 2));
    {
      ldm_handle::Scope sc(h, st, "attention", 4.0 * Bc * h->H * (double)h->S * h->S * h->dh, (double)M * (3 * HD + HD) * 2);
      launch_attention16(h->qkv16, h->att16, Bc, h->S, h->H, h->dh, 3 * HD, HD, st);
    }
    gemm("gemm_attn_out", 1, h->att16, HD, HD, f.w_out, HD, D, w.b_out, 0, h->P, h->Q, D, nullptr, 0,
         gemm_flops(M, D, D), (double)M * (HD * 2 + D * 8));
    {
      LnArgs a{};
      a.x = h->Q; a.p0 = w.g2; a.p1 = w.be2; a.y16 = h->h16;
      a.M = M; a.D = D; a.S = h->S; a.ld16 = Dq; a.ada = 0;
      ldm_handle::Scope sc(h, st, "layernorm2", 0, (double)M * D * 6);
      launch_layernorm(a, st);
    }
    gemm("gemm_ffn1", 2, h->h16, Dq, D, f.w1, Dq, F, w.b1, 1, nullptr, nullptr, 0, h->hid16, Fq,
         gemm_flops(M, F, D), (double)M * (D * 2 + F * 2));
    gemm("gemm_ffn2", 3, h->hid16, Fq, F, f.w2, Fq, D, w.b2, 0, h->Q, h->P, D, nullptr, 0,
         gemm_flops(M, D, F), (double)M * (F * 2 + D * 8));
  }
  {
    LnArgs a{};
    a.x = h->P; a.p0 = h->head_g; a.p1 = h->head_b; a.y16 = h->h16;
    a.M = M; a.D = D; a.S = h->S; a.ld16 = Dq; a.ada = 0;
    ldm_handle::Scope sc(h, st, "layernorm_head", 0, (double)M * D * 6);
    launch_layernorm(a, st);
  }
  gemm("gemm_head", 4, h->h16, Dq, D, h->fast_head, Dq, h->Cp, nullptr, 0, nullptr, h->logits, h->Cp,
       nullptr, 0, gemm_flops(M, C, D), (double)M * (D * 2 + C * 4));
  return 0;
}

// denoiser forward for `Bc` layouts whose tokens start at d_tokens -> h->logits [Bc*S, Cp]
static int denoise_chunk(ldm_handle* h, const int32_t* d_tokens, int t, int Bc, hipStream_t st, bool skip_embed = false) {
  if (h->cfg.precision == LDM_PREC_FAST_F16) return denoise_chunk_fast(h, d_tokens, t, Bc, st, skip_embed);
  const int M = Bc * h->S, D = h->D, F = h->F, C = h->C, Dp = h->Dp, Fp = h->Fp;
  const int prec = h->cfg.precision;
  const bool f16 = prec != LDM_PREC_EXACT_F32;
  const bool split = prec == LDM_PREC_SPLIT_F16;
  const size_t esz = f16 ? 2 : 4;
  for (int i = 0; i < h->L; ++i) {
    const LayerW& w = h->layers[i];
    const float* ss = h->adaln + ((size_t)t * h->L + i) * 2 * D;
    {  // AdaLN (layer 0: fused with the embedding gather); P <- normed x (the residual base)
      LnArgs a{};
      a.x = h->P;
      a.tokens = (i == 0) ? d_tokens : nullptr;
      a.emb = h->emb;
      a.pos = h->pos;
      a.p0 = ss;
      a.p1 = ss + D;
      a.y32 = h->P;
      a.y16 = f16 ? h->a16 : nullptr;
      a.y16lo = split ? h->a16lo : nullptr;
      a.M = M; a.D = D; a.S = h->S; a.ld16 = Dp; a.ada = 1;
      ldm_handle::Scope sc(h, st, i == 0 ? "embed_adaln" : "adaln", 0, (double)M * D * (4 + 4 + (f16 ? 2 : 0)));
      launch_layernorm(a, st);
    }
    {  // QKV projection
      GemmArgs g{};
      g.A = f16 ? (const void*)h->a16 : (const void*)h->P;
      g.Alo = h->a16lo;
      g.W = f16 ? (const void*)w.w_in16 : (const void*)w.w_in;
      g.Wlo = w.w_in16lo;
      g.bias = w.b_in;
      g.C32 = (prec == LDM_PREC_FAST_F16) ? nullptr : h->qkv32;
      g.C16 = (prec == LDM_PREC_FAST_F16) ? h->qkv16 : nullptr;
      g.M = M; g.N = 3 * D; g.K = f16 ? Dp : D; g.lda = f16 ? Dp : D; g.ldw = f16 ? Dp : D;
      g.ldc32 = 3 * D; g.ldc16 = 3 * D; g.precision = prec;
      ldm_handle::Scope sc(h, st, "gemm_qkv", gemm_flops(M, 3 * D, D), (double)M * D * esz + (double)M * 3 * D * (prec == 1 ? 2 : 4));
      launch_gemm(g, st);
    }
    {  // attention
      AttnArgs a{};
      a.in_f16 = (prec == LDM_PREC_FAST_F16);
      a.qkv = a.in_f16 ? (const void*)h->qkv16 : (const void*)h->qkv32;
      a.out32 = f16 ? nullptr : h->att32;
      a.out16 = f16 ? h->att16 : nullptr;
      a.out16lo = split ? h->att16lo : nullptr;
      a.B = Bc; a.S = h->S; a.H = h->H; a.dh = h->dh; a.D = D; a.ld = 3 * D; a.ldo32 = D; a.ldo16 = Dp;
      ldm_handle::Scope sc(h, st, "attention", 4.0 * Bc * h->H * (double)h->S * h->S * h->dh,
                           (double)M * 3 * D * (a.in_f16 ? 2 : 4) + (double)M * D * esz);
      launch_attention(a, st);
    }
    {  // out-proj + residual onto the normed x:  Q = P + att·Wo^T + bo
      GemmArgs g{};
      g.A = f16 ? (const void*)h->att16 : (const void*)h->att32;
      g.Alo = h->att16lo;
      g.W = f16 ? (const void*)w.w_out16 : (const void*)w.w_out;
      g.Wlo = w.w_out16lo;
      g.bias = w.b_out;
      g.res = h->P; g.ldres = D;
      g.C32 = h->Q; g.ldc32 = D;
      g.M = M; g.N = D; g.K = f16 ? Dp : D; g.lda = f16 ? Dp : D; g.ldw = f16 ? Dp : D; g.precision = prec;
      ldm_handle::Scope sc(h, st, "gemm_attn_out", gemm_flops(M, D, D), (double)M * D * (esz + 8));
      launch_gemm(g, st);
    }
    {  // LayerNorm 2
      LnArgs a{};
      a.x = h->Q; a.p0 = w.g2; a.p1 = w.be2;
      a.y32 = f16 ? nullptr : h->h32;
      a.y16 = f16 ? h->h16 : nullptr;
      a.y16lo = split ? h->h16lo : nullptr;
      a.M = M; a.D = D; a.S = h->S; a.ld16 = Dp; a.ada = 0;
      ldm_handle::Scope sc(h, st, "layernorm2", 0, (double)M * D * (4 + esz));
      launch_layernorm(a, st);
    }
    {  // FFN1 + ReLU
      GemmArgs g{};
      g.A = f16 ? (const void*)h->h16 : (const void*)h->h32;
      g.Alo = h->h16lo;
      g.W = f16 ? (const void*)w.w1_16 : (const void*)w.w1;
      g.Wlo = w.w1_16lo;
      g.bias = w.b1; g.relu = 1;
      g.C32 = f16 ? nullptr : h->hid32; g.ldc32 = F;
      g.C16 = f16 ? h->hid16 : nullptr; g.C16lo = split ? h->hid16lo : nullptr; g.ldc16 = Fp;
      g.M = M; g.N = F; g.K = f16 ? Dp : D; g.lda = f16 ? Dp : D; g.ldw = f16 ? Dp : D; g.precision = prec;
      ldm_handle::Scope sc(h, st, "gemm_ffn1", gemm_flops(M, F, D), (double)M * D * esz + (double)M * F * esz);
      launch_gemm(g, st);
    }
    {  // FFN2 + residual:  P = Q + hid·W2^T + b2
      GemmArgs g{};
      g.A = f16 ? (const void*)h->hid16 : (const void*)h->hid32;
      g.Alo = h->hid16lo;
      g.W = f16 ? (const void*)w.w2_16 : (const void*)w.w2;
      g.Wlo = w.w2_16lo;
      g.bias = w.b2;
      g.res = h->Q; g.ldres = D;
      g.C32 = h->P; g.ldc32 = D;
      g.M = M; g.N = D; g.K = f16 ? Fp : F; g.lda = f16 ? Fp : F; g.ldw = f16 ? Fp : F; g.precision = prec;
      ldm_handle::Scope sc(h, st, "gemm_ffn2", gemm_flops(M, D, F), (double)M * F * esz + (double)M * D * 8);
      launch_gemm(g, st);
    }
  }
  {  // head: LayerNorm + vocab projection (no bias)
    LnArgs a{};
    a.x = h->P; a.p0 = h->head_g; a.p1 = h->head_b;
    a.y32 = f16 ? nullptr : h->h32;
    a.y16 = f16 ? h->h16 : nullptr;
    a.y16lo = split ? h->h16lo : nullptr;
    a.M = M; a.D = D; a.S = h->S; a.ld16 = Dp; a.ada = 0;
    {
      ldm_handle::Scope sc(h, st, "layernorm_head", 0, (double)M * D * (4 + esz));
      launch_layernorm(a, st);
    }
    GemmArgs g{};
    g.A = f16 ? (const void*)h->h16 : (const void*)h->h32;
    g.Alo = h->h16lo;
    g.W = f16 ? (const void*)h->head_w16 : (const void*)h->head_w;
    g.Wlo = h->head_w16lo;
    g.C32 = h->logits; g.ldc32 = h->Cp;
    g.M = M; g.N = C; g.K = f16 ? Dp : D; g.lda = f16 ? Dp : D; g.ldw = f16 ? Dp : D; g.precision = prec;
    ldm_handle::Scope sc(h, st, "gemm_head", gemm_flops(M, C, D), (double)M * D * esz + (double)M * C * 4);
    launch_gemm(g, st);
  }
  return 0;
}

static void fill_post(ldm_handle* h, PostArgs& p, const ldm_cond* cond, const ldm_sampler* s, size_t layout_off,
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

static int check_ready(ldm_handle* h, int B) {
  if (!h) return -1;
  h->activate(0);
  if (!h->finalized) return h->fail(-5, "weights not finalized: call ldm_finalize_weights first");
  if (B < 1 || B > h->cfg.max_batch) return h->fail(-1, "batch %d outside [1, max_batch=%d]", B, h->cfg.max_batch);
  return 0;
}

static int check_sampler(ldm_handle* h, const ldm_sampler* s) {
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
  for (int l = 0; l < h->n_lanes; ++l) {
    if (h->ws[l].rel_logp) continue;
    float* buf = nullptr;
    int rc = h->dalloc(&buf, (size_t)h->chunk * h->C * h->S, false);
    if (rc) return rc;
    h->ws[l].rel_logp = buf;
    if (l == h->cur_lane) h->rel_logp = buf;
  }
  return 0;
}

static void fill_rel(ldm_handle* h, RelArgs& a, const ldm_relation* rel, size_t layout_off, int Bc) {
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
static bool loop_fusable(const ldm_handle* h, const ldm_relation* rel) {
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

static int set_rng(ldm_handle* h, uint64_t seed, uint64_t first_layout, hipStream_t st) {
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
  const int first = lane < 0 ? 0 : lane * h->chunk;
  const int stride = lane < 0 ? h->chunk : h->n_lanes * h->chunk;
  h->activate(lane < 0 ? 0 : lane);
  for (int off = first; off < B; off += stride) {
    const int Bc = std::min(h->chunk, B - off);
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

// ------------------------------------------------------------------------------------------ introspection
// "key=value;..." description of what this handle runs: numerics mode, kernel family, chunk / lanes, near-tie thresholds
// and the development knobs the library has honoured in this process (ldm_knobs.h).  Returns the length needed.
extern "C" int ldm_describe(const ldm_handle* h, char* buf, int cap) {
  if (!h) return -1;
  static const char* prec[3] = {"exact_f32", "fast_f16", "split_f16"};
  const bool loop = loop_fusable(h, nullptr);
  char tmp[768];
  const int n = snprintf(tmp, sizeof(tmp),
                         "abi=%d;precision=%s;kernels=%s;loop=%s;chunk=%d;lanes=%d;lane_offset_us=%d;tie_rel=%g;tie_abs=%g;knobs=%s",
                         LDM_ABI_VERSION, prec[h->cfg.precision],
                         h->cfg.precision != LDM_PREC_FAST_F16 ? "tiled_gemm+attn" : h->fused_attn == 6 ? "stack" : "generic16",
                         loop ? "one_launch" : "per_step_graph", h->chunk, h->n_lanes, h->lane_offset_us, (double)h->tie_rel,
                         (double)h->tie_abs, knobs_honoured().c_str());
  if (buf && cap > 0) {
    strncpy(buf, tmp, (size_t)cap - 1);
    buf[cap - 1] = 0;
  }
  return n;
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
