// C-ABI of the FID feature extractor (include/ldm_hip.h, section "FID feature extractor"): owns the transposed
// weights of FIDNetV3's encoder half and launches kernels_fid.hip.  Separate handle type: the FID network is
// independent of the diffusion model (its own checkpoint: trainer/fid/model.py:182-193).
#include "../../include/ldm_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ldm_kernels.h"

using namespace ldm;

static thread_local std::string g_fid_create_error;

struct ldm_fid {
  int device = 0, num_label = 0, max_bbox = 0, n_layer = 0;
  std::string err;
  std::map<std::string, std::pair<std::vector<float>, std::vector<int64_t>>> raw;  // host copies until finalize
  std::vector<void*> owned;
  FidArgs args{};
  bool finalized = false;
  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
};

namespace {
struct Dev {
  int prev = -1;
  bool ok = true;
  explicit Dev(int d) {
    if (hipGetDevice(&prev) != hipSuccess) ok = false;
    else if (prev != d) ok = hipSetDevice(d) == hipSuccess;
    else prev = -1;
  }
  ~Dev() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
}  // namespace

extern "C" const char* ldm_fid_last_error(const ldm_fid* h) { return h ? h->err.c_str() : g_fid_create_error.c_str(); }

extern "C" int ldm_fid_create(int num_label, int max_bbox, int d_model, int n_head, int n_layer, int device, ldm_fid** out) {
  auto bad = [&](const char* m) {
    g_fid_create_error = m;
    return -1;
  };
  if (!out) return bad("null argument");
  if (d_model != 256 || n_head != 4) return bad("FIDNetV3 geometry: d_model 256, 4 heads (trainer/fid/model.py:124)");
  if (n_layer < 1 || n_layer > 8) return bad("n_layer must be in [1, 8]");
  if (max_bbox < 1 || max_bbox > 31) return bad("max_bbox must be in [1, 31] (one workgroup holds token + elements in LDS)");
  if (num_label < 1) return bad("num_label must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bad("no HIP device visible: the MI355X path has no CPU fallback");
  if (device < 0 || device >= ndev) return bad("device index out of range");
  auto* h = new ldm_fid();
  h->device = device;
  h->num_label = num_label;
  h->max_bbox = max_bbox;
  h->n_layer = n_layer;
  *out = h;
  return 0;
}

extern "C" void ldm_fid_destroy(ldm_fid* h) {
  if (!h) return;
  Dev g(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->owned) (void)hipFree(p);
  delete h;
}

// key = FIDNetV3 state_dict key (trainer/fid/model.py:127-151); decoder-half keys are accepted and ignored
extern "C" int ldm_fid_load_weight(ldm_fid* h, const char* key, const float* h_data, const int64_t* shape, int ndim) {
  if (!h || !key || !h_data || (ndim > 0 && !shape)) return h ? h->fail(-1, "null argument") : -1;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= shape[i];
  if (n <= 0) return h->fail(-1, "empty tensor for key %s", key);
  std::string k(key);
  for (const char* p : {"module.", "model."})
    if (k.compare(0, strlen(p), p) == 0) k = k.substr(strlen(p));
  h->raw[k] = {std::vector<float>(h_data, h_data + n), std::vector<int64_t>(shape, shape + ndim)};
  h->finalized = false;
  return 0;
}

static int upload(ldm_fid* h, const std::vector<float>& v, const float** out) {
  void* d = nullptr;
  if (hipMalloc(&d, v.size() * 4) != hipSuccess) return h->fail(-3, "hipMalloc failed");
  h->owned.push_back(d);
  if (hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return h->fail(-2, "hipMemcpy failed");
  *out = (const float*)d;
  return 0;
}
// checkpoint tensor [N][K] (or any shape with N*K elements when transpose == false) -> device; transposed to [K][N]
static int take(ldm_fid* h, const std::string& key, std::initializer_list<int64_t> shape, bool transpose, const float** out) {
  auto it = h->raw.find(key);
  if (it == h->raw.end()) return h->fail(-4, "missing FIDNetV3 checkpoint key: %s", key.c_str());
  if (it->second.second != std::vector<int64_t>(shape)) return h->fail(-4, "FIDNetV3 key %s has an unexpected shape", key.c_str());
  const std::vector<float>& src = it->second.first;
  if (!transpose) return upload(h, src, out);
  const int64_t N = *shape.begin(), K = *(shape.begin() + 1);
  std::vector<float> t((size_t)N * K);
  for (int64_t n = 0; n < N; ++n)
    for (int64_t k = 0; k < K; ++k) t[(size_t)k * N + n] = src[(size_t)n * K + k];
  return upload(h, t, out);
}

extern "C" int ldm_fid_finalize(ldm_fid* h) {
  if (!h) return -1;
  Dev g(h->device);
  if (!g.ok) return h->fail(-2, "hipSetDevice failed");
  for (void* p : h->owned) (void)hipFree(p);
  h->owned.clear();
  FidArgs& a = h->args;
  a = FidArgs{};
  int rc;
  const int D = 256;
  if ((rc = take(h, "emb_label.weight", {h->num_label, D}, false, &a.emb_label))) return rc;
  if ((rc = take(h, "fc_bbox.weight", {D, 4}, true, &a.fc_bbox_wt))) return rc;
  if ((rc = take(h, "fc_bbox.bias", {D}, false, &a.fc_bbox_b))) return rc;
  if ((rc = take(h, "enc_fc_in.weight", {D, 2 * D}, true, &a.fc_in_wt))) return rc;
  if ((rc = take(h, "enc_fc_in.bias", {D}, false, &a.fc_in_b))) return rc;
  if ((rc = take(h, "enc_transformer.token", {1, 1, D}, false, &a.token))) return rc;
  for (int i = 0; i < h->n_layer; ++i) {
    const std::string b = "enc_transformer.core.layers." + std::to_string(i) + ".";
    FidLayer& L = a.layer[i];
    if ((rc = take(h, b + "self_attn.in_proj_weight", {3 * D, D}, true, &L.in_wt))) return rc;
    if ((rc = take(h, b + "self_attn.in_proj_bias", {3 * D}, false, &L.in_b))) return rc;
    if ((rc = take(h, b + "self_attn.out_proj.weight", {D, D}, true, &L.out_wt))) return rc;
    if ((rc = take(h, b + "self_attn.out_proj.bias", {D}, false, &L.out_b))) return rc;
    if ((rc = take(h, b + "linear1.weight", {D / 2, D}, true, &L.w1t))) return rc;
    if ((rc = take(h, b + "linear1.bias", {D / 2}, false, &L.b1))) return rc;
    if ((rc = take(h, b + "linear2.weight", {D, D / 2}, true, &L.w2t))) return rc;
    if ((rc = take(h, b + "linear2.bias", {D}, false, &L.b2))) return rc;
    if ((rc = take(h, b + "norm1.weight", {D}, false, &L.n1_g))) return rc;
    if ((rc = take(h, b + "norm1.bias", {D}, false, &L.n1_b))) return rc;
    if ((rc = take(h, b + "norm2.weight", {D}, false, &L.n2_g))) return rc;
    if ((rc = take(h, b + "norm2.bias", {D}, false, &L.n2_b))) return rc;
  }
  a.n_layer = h->n_layer;
  a.num_label = h->num_label;
  h->finalized = true;
  return 0;
}

extern "C" int ldm_fid_features(ldm_fid* h, const float* d_bbox, const int64_t* d_label, const uint8_t* d_padding_mask,
                                int B, int N, float* d_feat, void* stream) {
  if (!h) return -1;
  if (!h->finalized) return h->fail(-5, "weights not finalized: call ldm_fid_finalize first");
  if (B < 0 || N < 1 || N > h->max_bbox) return h->fail(-1, "B must be >= 0 and N in [1, max_bbox=%d]", h->max_bbox);
  if (B == 0) return 0;
  if (!d_bbox || !d_label || !d_padding_mask || !d_feat) return h->fail(-1, "null argument");
  Dev g(h->device);
  if (!g.ok) return h->fail(-2, "hipSetDevice failed");
  // a stale error of an unrelated earlier runtime call (e.g. the caller's framework probing a host pointer with
  // hipPointerGetAttributes) must not be reported as this launch's
  (void)hipGetLastError();
  FidArgs a = h->args;
  a.bbox = d_bbox;
  a.label = d_label;
  a.padding_mask = d_padding_mask;
  a.feat = d_feat;
  a.N = N;
  launch_fid_features(a, B, (hipStream_t)stream);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return h->fail(-2, "fid_features_k launch failed: %s", hipGetErrorString(e));
  return 0;
}

// ------------------------------------------------------------------------------------------ PRDC
extern "C" int ldm_prdc(const float* d_real, int n_real, const float* d_fake, int n_fake, int dim, int nearest_k,
                        float* h_out4, void* stream) {
  if (!d_real || !d_fake || !h_out4 || n_real < 1 || n_fake < 1 || dim < 1) return -1;
  if (nearest_k < 1 || nearest_k > 7 || nearest_k + 1 > n_real || nearest_k + 1 > n_fake) return -1;
  hipStream_t st = (hipStream_t)stream;
  const size_t nmax = (size_t)std::max(n_real, n_fake);
  // the pairwise-distance workspace is nmax^2 floats (prdc itself builds the same matrices): refuse sets whose matrix would
  // not fit comfortably (> 16 GiB: 65 536 features per set) instead of failing opaquely inside hipMalloc
  if (nmax > 65536) return -6;
  float *D = nullptr, *r2r = nullptr, *r2f = nullptr;
  unsigned long long* cnt = nullptr;
  auto done = [&](int rc) {
    if (D) (void)hipFree(D);
    if (r2r) (void)hipFree(r2r);
    if (r2f) (void)hipFree(r2f);
    if (cnt) (void)hipFree(cnt);
    return rc;
  };
  if (hipMalloc((void**)&D, nmax * nmax * sizeof(float)) != hipSuccess) return done(-3);
  if (hipMalloc((void**)&r2r, n_real * sizeof(float)) != hipSuccess) return done(-3);
  if (hipMalloc((void**)&r2f, n_fake * sizeof(float)) != hipSuccess) return done(-3);
  if (hipMalloc((void**)&cnt, 4 * sizeof(unsigned long long)) != hipSuccess) return done(-3);
  if (hipMemsetAsync(cnt, 0, 4 * sizeof(unsigned long long), st) != hipSuccess) return done(-2);
  launch_prdc_pdist2(d_real, n_real, d_real, n_real, dim, D, st);
  launch_prdc_kth(D, n_real, n_real, nearest_k + 1, r2r, st);
  launch_prdc_pdist2(d_fake, n_fake, d_fake, n_fake, dim, D, st);
  launch_prdc_kth(D, n_fake, n_fake, nearest_k + 1, r2f, st);
  launch_prdc_pdist2(d_real, n_real, d_fake, n_fake, dim, D, st);
  launch_prdc_counts(D, n_real, n_fake, r2r, r2f, cnt, st);
  unsigned long long h[4];
  if (hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess) return done(-2);
  if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return done(-2);
  // (the ratios in double like prdc, narrowed at the ABI)
  h_out4[0] = (float)((double)h[0] / (double)n_fake);                           // precision
  h_out4[1] = (float)((double)h[1] / (double)n_real);                           // recall
  h_out4[2] = (float)((double)h[2] / ((double)nearest_k * (double)n_fake));     // density
  h_out4[3] = (float)((double)h[3] / (double)n_real);                           // coverage
  return done(0);
}

// ------------------------------------------------------------------------------------------ alignment / overlap
// compute_alignment + compute_overlap (trainer/helpers/metric.py:98-203) on decoded layouts resident in HBM: six scores per layout
extern "C" int ldm_layout_metrics(const float* d_bbox, const uint8_t* d_mask, int B, int S, float* d_out6, void* stream) {
  if (B < 0 || (B > 0 && (!d_bbox || !d_mask || !d_out6))) return -1;
  if (B == 0) return 0;
  (void)hipGetLastError();
  if (launch_layout_metrics(d_bbox, d_mask, B, S, d_out6, (hipStream_t)stream)) return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
