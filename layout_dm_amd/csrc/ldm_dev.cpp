// Development hooks of libldm_hip.so (NOT part of the public ABI in include/ldm_hip.h): micro-benchmarks on synthetic
// operands for tools/gemm_tune.py / ffn_quick.py / attn_bench.py and the s_memtime phase sums of the instrumented kernel
// variants for tools/phase_probe.py.  Nothing in the product path calls into this file.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "ldm_kernels.h"
#include "ldm_pack.h"

using namespace ldm;
using ldm_pack::pack_ffn_image;

namespace {

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

uint16_t f2h_bits(float x) {
  const __half hh = __float2half(x);
  uint16_t u;
  memcpy(&u, &hh, 2);
  return u;
}

// device buffers / events of one benchmark call, released on every return path
struct DevScope {
  std::vector<void*> bufs;
  hipEvent_t a = nullptr, b = nullptr;
  template <typename T>
  bool alloc(T** p, size_t bytes, const void* host = nullptr) {
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return false;
    bufs.push_back(d);
    *p = static_cast<T*>(d);
    return (host ? hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) : hipMemset(d, 0, bytes)) == hipSuccess;
  }
  bool events() { return hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess; }
  // average ms per call of `run` over `iters` calls (3 untimed warm-up calls); < 0 on a HIP error
  template <typename F>
  float time(int iters, F run) {
    for (int i = 0; i < 3; ++i) run();
    if (hipEventRecord(a, 0) != hipSuccess) return -1.f;
    for (int i = 0; i < iters; ++i) run();
    float ms = 0;
    if (hipEventRecord(b, 0) != hipSuccess || hipEventSynchronize(b) != hipSuccess ||
        hipEventElapsedTime(&ms, a, b) != hipSuccess || hipGetLastError() != hipSuccess)
      return -1.f;
    return ms / iters;
  }
  ~DevScope() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    for (void* p : bufs) (void)hipFree(p);
  }
};

}  // namespace

// Tile-configuration tuning aid: times launch_gemm16 (cfg < 100), the row-stationary GEMM (cfg 100) or the two-launch
// fused FFN (cfg 101: A = [M,512] LN output, N = d_model 464, hidden 1856) on synthetic operands.  Average ms per launch.
extern "C" int ldm_dev_bench_gemm(int M, int N, int K, int cfg, int iters, float* ms_out) {
  const int Mp = round_up(M, 256), Np = round_up(N, 256), Kp = round_up(K, 64);
  std::vector<uint16_t> ha((size_t)Mp * Kp), hw((size_t)Np * Kp);
  uint32_t s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return ((float)(s >> 8) / 8388608.0f) - 1.0f;
  };
  for (auto& x : ha) x = f2h_bits(rnd());
  for (auto& x : hw) x = f2h_bits(rnd() * 0.05f);
  DevScope d;
  __half *A = nullptr, *W = nullptr, *Cc = nullptr, *W1b = nullptr;
  float *bias = nullptr, *res = nullptr, *out32 = nullptr, *bias1 = nullptr;
  if (!d.alloc(&A, ha.size() * 2, ha.data()) || !d.alloc(&W, hw.size() * 2, hw.data()) ||
      !d.alloc(&Cc, (size_t)Mp * Np * 2) || !d.alloc(&bias, (size_t)Np * 4) || !d.events())
    return -3;
  if (cfg == 101) {
    std::vector<uint16_t> h1((size_t)2048 * 512), h2((size_t)512 * 1856);
    for (auto& x : h1) x = f2h_bits(rnd() * 0.05f);
    for (auto& x : h2) x = f2h_bits(rnd() * 0.05f);
    const std::vector<uint16_t> img = pack_ffn_image(h1.data(), h2.data(), 1856, 1856, 480);
    if (!d.alloc(&W1b, img.size() * 2, img.data()) || !d.alloc(&bias1, 2048 * 4) ||
        !d.alloc(&res, (size_t)Mp * N * 4) || !d.alloc(&out32, (size_t)Mp * N * 4))
      return -3;
  }
  GemmArgs g{};
  g.A = A; g.W = W; g.C16 = Cc; g.ldc16 = Np; g.M = M; g.N = N; g.K = round_up(K, gemm16_block_k(cfg));
  g.lda = Kp; g.ldw = Kp; g.precision = 1; g.bias = bias; g.relu = 1;
  const float ms = d.time(iters, [&]() {
    if (cfg == 100) {
      GemmArgs r = g;
      r.K = K;
      r.relu = 0;
      launch_rowgemm(r, 0, nullptr, 0);
    } else if (cfg == 101) {
      launch_ffn_fused(A, Kp, W1b, nullptr, bias1, bias, res, out32, N, M, N, 1856, nullptr, nullptr, 0);
    } else {
      launch_gemm16(g, cfg, 2, 0);
    }
  });
  *ms_out = ms;
  return ms >= 0.f ? 0 : -2;
}

// attention micro-benchmark: B layouts x 8 heads on random fp16 qkv (the stand-alone attn_mfma_k)
extern "C" int ldm_dev_bench_attn(int B, int iters, float* ms_out) {
  const int S = 125, H = 8, ldq = 3 * H * 64, ldo = H * 64;
  const size_t rows = (size_t)B * S + 256;
  std::vector<uint16_t> hq(rows * ldq);
  uint32_t s = 777u;
  for (auto& x : hq) {
    s = s * 1664525u + 1013904223u;
    x = f2h_bits(((float)(s >> 8) / 8388608.0f) - 1.0f);
  }
  DevScope d;
  __half *q = nullptr, *o = nullptr;
  if (!d.alloc(&q, hq.size() * 2, hq.data()) || !d.alloc(&o, rows * ldo * 2) || !d.events()) return -3;
  const float ms = d.time(iters, [&]() { launch_attention16(q, o, B, S, H, 58, ldq, ldo, 0); });
  *ms_out = ms;
  return ms >= 0.f ? 0 : -2;
}

// s_memtime phase sums of the instrumented kernel variants (LDM_FFN_DBG=3 / LDM_ATTN_TM=1 / LDM_LAYER_TM / LDM_STACK_TM);
// every read resets the counters
namespace ldm {
void ffn_phase_read(unsigned long long* out12);
void attn_phase_read(unsigned long long* out16);
void layer_phase_read(unsigned long long* out16);
void stack_phase_read(unsigned long long* out16);
}  // namespace ldm
extern "C" void ldm_dev_ffn_phases(unsigned long long* out12) { ldm::ffn_phase_read(out12); }
extern "C" void ldm_dev_attn_phases(unsigned long long* out16) { ldm::attn_phase_read(out16); }
extern "C" void ldm_dev_layer_phases(unsigned long long* out16) { ldm::layer_phase_read(out16); }
extern "C" void ldm_dev_stack_phases(unsigned long long* out16) { ldm::stack_phase_read(out16); }
