// Development hooks of libldm_hip.so (NOT part of the public ABI in include/ldm_hip.h): micro-benchmarks on synthetic
// operands for tools/gemm_tune.py and the s_memtime phase sums of the instrumented stack kernel for tools/phase_probe.py.  Nothing in the product path calls into this file.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "ldm_kernels.h"

using namespace ldm;

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

// Tile-configuration tuning aid of the generic fast path: times launch_gemm16 on synthetic operands.  Average ms per launch.
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
  __half *A = nullptr, *W = nullptr, *Cc = nullptr;
  float* bias = nullptr;
  if (!d.alloc(&A, ha.size() * 2, ha.data()) || !d.alloc(&W, hw.size() * 2, hw.data()) ||
      !d.alloc(&Cc, (size_t)Mp * Np * 2) || !d.alloc(&bias, (size_t)Np * 4) || !d.events())
    return -3;
  GemmArgs g{};
  g.A = A; g.W = W; g.C16 = Cc; g.ldc16 = Np; g.M = M; g.N = N; g.K = round_up(K, gemm16_block_k(cfg));
  g.lda = Kp; g.ldw = Kp; g.precision = 1; g.bias = bias; g.relu = 1;
  const float ms = d.time(iters, [&]() { launch_gemm16(g, cfg, 2, 0); });
  *ms_out = ms;
  return ms >= 0.f ? 0 : -2;
}

// Split GEMM (gemm16x3_k, 256 x 128 tiles) on synthetic operands: abl = 0 the kernel, 1 operand fills only, 2 fragment reads +
// MFMAs only, 3 fills + MFMAs without fragment reads (tools/gemm_x3_probe.py).  Average ms per launch.
extern "C" int ldm_dev_bench_gemm_x3(int M, int N, int K, int abl, int c16, int iters, float* ms_out) {
  const int Mp = round_up(M, 256), Np = round_up(N, 256), Kp = round_up(K, 64);
  std::vector<uint16_t> ha((size_t)Mp * Kp), hw((size_t)Np * Kp);
  uint32_t s = 4321u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return ((float)(s >> 8) / 8388608.0f) - 1.0f;
  };
  for (auto& x : ha) x = f2h_bits(rnd());
  for (auto& x : hw) x = f2h_bits(rnd() * 0.05f);
  DevScope d;
  __half *A = nullptr, *Al = nullptr, *W = nullptr, *Wl = nullptr, *C16 = nullptr, *C16l = nullptr;
  float *C32 = nullptr, *bias = nullptr;
  if (!d.alloc(&A, ha.size() * 2, ha.data()) || !d.alloc(&Al, ha.size() * 2, ha.data()) || !d.alloc(&W, hw.size() * 2, hw.data()) ||
      !d.alloc(&Wl, hw.size() * 2, hw.data()) || !d.alloc(&C32, (size_t)Mp * Np * 4) || !d.alloc(&C16, (size_t)Mp * Np * 2) ||
      !d.alloc(&C16l, (size_t)Mp * Np * 2) || !d.alloc(&bias, (size_t)Np * 4) || !d.events())
    return -3;
  GemmArgs g{};
  g.A = A; g.Alo = Al; g.W = W; g.Wlo = Wl; g.M = M; g.N = N; g.K = Kp; g.lda = Kp; g.ldw = Kp; g.precision = 2; g.bias = bias;
  if (c16) { g.C16 = C16; g.C16lo = C16l; g.ldc16 = Np; g.relu = 1; } else { g.C32 = C32; g.ldc32 = Np; }
  const float ms = d.time(iters, [&]() { launch_gemm16x3_abl(g, abl, 0); });
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

// s_memtime phase sums of the instrumented stack kernel (LDM_ATTN_TM=1, per-step path); every read resets the counters
namespace ldm {
void stack_phase_read(unsigned long long* out16);
}  // namespace ldm
extern "C" void ldm_dev_stack_phases(unsigned long long* out16) { ldm::stack_phase_read(out16); }
extern "C" void ldm_dev_lngemm_phases(unsigned long long* out8) { ldm::lngemm_phase_read(out8); }
