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
extern "C" void ldm_dev_attnout_phases(unsigned long long* out24) { ldm::attnout_phase_read(out24); }

// Unit check of the split mode's fused attention + out_proj launch (kernels_attnout.hip) on synthetic operands against a float64
// host computation of the same block: q / k / v (amplitude qk_amp / 1) -> hi / lo panels as in_proj's epilogue writes them, out_proj
// weights with max |w| in [1, 2) (what ldm_weights.cpp make_w16's power-of-two pre-scale produces) -> k-step image, residual rows,
// bias.  err_out[0] = max |out - ref| / max |ref - res| over the B layouts (every row, every column), err_out[1] = the same for a
// computation from the fp16 hi parts only (what a dropped lo term would look like: the yardstick), err_out[2] = max |ref - res|.
// tests/test_attnout_gpu.py.
#include <cmath>

#include "ldm_pack.h"
// zero_lo (debugging aid): bit 0 / 1 / 2 / 3 = drop the lo halves of q / k / v / the weights from the INPUTS (kernel and reference alike)
extern "C" int ldm_dev_attnout_check(int B, int S, float qk_amp, uint32_t seed, double* err_out, int zero_lo) {
  const int H = 8, dh = 58, D = 464, NP = 48;
  if (B < 1 || S < 1 || S > 128) return -1;
  const size_t M = (size_t)B * S, rows = M + 128;   // (a layout reads 128 rows from its first one)
  const size_t PS = rows * 64;
  uint32_t s = seed * 2654435761u + 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return ((float)(s >> 8) / 8388608.0f) - 1.0f;
  };
  std::vector<float> q(M * H * dh), k(M * H * dh), v(M * H * dh), res(M * D), bias(D), wo((size_t)D * D);
  for (auto& x : q) x = rnd() * qk_amp;
  for (auto& x : k) x = rnd() * qk_amp;
  for (auto& x : v) x = rnd();
  for (auto& x : res) x = rnd() * 2.0f;
  for (auto& x : bias) x = rnd() * 0.1f;
  for (auto& x : wo) x = rnd() * 1.9f;
  if (zero_lo & 16)   // (debug: W = 32 I, so that out - res - bias IS the attention output)
    for (int n = 0; n < D; ++n)
      for (int kk = 0; kk < D; ++kk) wo[(size_t)n * D + kk] = n == kk ? 32.0f : 0.0f;
  const float out_scale = 1.0f / 32.0f;
  auto split = [&](float x, uint16_t& hi, uint16_t& lo) {
    const __half hh = __float2half(x);
    const __half ll = __float2half(x - __half2float(hh));
    memcpy(&hi, &hh, 2);
    memcpy(&lo, &ll, 2);
  };
  std::vector<uint16_t> ph((size_t)NP * rows * 32, 0), pl((size_t)NP * rows * 32, 0);
  const std::vector<float>* src[3] = {&q, &k, &v};
  for (int which = 0; which < 3; ++which)
    for (size_t r = 0; r < M; ++r)
      for (int h = 0; h < H; ++h)
        for (int d = 0; d < dh; ++d) {
          const size_t pn = (size_t)(which * H + h) * 2 + d / 32;
          split((*src[which])[(r * H + h) * dh + d], ph[(pn * rows + r) * 32 + d % 32], pl[(pn * rows + r) * 32 + d % 32]);
          if (zero_lo & (1 << which)) pl[(pn * rows + r) * 32 + d % 32] = 0;
        }
  std::vector<uint16_t> wh((size_t)D * 512, 0), wl((size_t)D * 512, 0);
  for (int n = 0; n < D; ++n)
    for (int kk = 0; kk < D; ++kk) {
      split(wo[(size_t)n * D + kk], wh[(size_t)n * 512 + kk], wl[(size_t)n * 512 + kk]);
      if (zero_lo & 8) wl[(size_t)n * 512 + kk] = 0;
    }
  const std::vector<uint16_t> img = ldm_pack::pack_x3_kstep_image(wh.data(), wl.data(), D, 512, H, dh);
  DevScope dv;
  char *dph = nullptr, *dpl = nullptr, *dimg = nullptr;
  float *dres = nullptr, *dbias = nullptr, *dout = nullptr;
  if (!dv.alloc(&dph, ph.size() * 2, ph.data()) || !dv.alloc(&dpl, pl.size() * 2, pl.data()) || !dv.alloc(&dimg, img.size() * 2, img.data()) ||
      !dv.alloc(&dres, res.size() * 4, res.data()) || !dv.alloc(&dbias, bias.size() * 4, bias.data()) || !dv.alloc(&dout, res.size() * 4))
    return -3;
  AttnOutArgs a{};
  a.qkv_hi = dph; a.qkv_lo = dpl; a.panel_stride = PS; a.w_img = dimg; a.res = dres; a.bias = dbias; a.out = dout;
  a.S = S; a.D = D; a.scale = 1.0f / sqrtf((float)dh); a.out_scale = out_scale;
  if (launch_attnout16x3(a, B, 0)) return -4;
  std::vector<float> out(res.size());
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess ||
      hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
    return -2;
  double e_full = 0, e_hi = 0, mag = 0;
  for (int i = 3; i < 25; ++i) err_out[i] = 0;   // (debug: [3..5] location of the maximum, [6..9] per wave, [10..24] per column tile)
  auto h2d = [&](uint16_t u) {
    __half hh;
    memcpy(&hh, &u, 2);
    return (double)__half2float(hh);
  };
  std::vector<double> att((size_t)S * H * dh), att_hi((size_t)S * H * dh), p(S);
  for (int b = 0; b < B; ++b) {
    for (int variant = 0; variant < 2; ++variant) {   // 0: full values, 1: fp16 hi parts only
      std::vector<double>& o = variant ? att_hi : att;
      auto val = [&](int which, size_t r, int h, int d) {
        const size_t pn = (size_t)(which * H + h) * 2 + d / 32, i = (pn * rows + r) * 32 + d % 32;
        return variant ? h2d(ph[i]) : h2d(ph[i]) + h2d(pl[i]);
      };
      for (int h = 0; h < H; ++h)
        for (int i = 0; i < S; ++i) {
          double mx = -1e300;
          for (int j = 0; j < S; ++j) {
            double sc = 0;
            for (int d = 0; d < dh; ++d) sc += val(0, (size_t)b * S + i, h, d) * val(1, (size_t)b * S + j, h, d);
            p[j] = sc * (double)a.scale;
            mx = std::max(mx, p[j]);
          }
          double sum = 0;
          for (int j = 0; j < S; ++j) sum += (p[j] = std::exp(p[j] - mx));
          for (int d = 0; d < dh; ++d) {
            double acc = 0;
            for (int j = 0; j < S; ++j) acc += p[j] * val(2, (size_t)b * S + j, h, d);
            o[((size_t)i * H + h) * dh + d] = acc / sum;
          }
        }
    }
    for (int i = 0; i < S; ++i)
      for (int n = 0; n < D; ++n) {
        double acc = 0, acc_hi = 0;
        for (int kk = 0; kk < D; ++kk) {
          acc += att[(size_t)i * D + kk] * (h2d(wh[(size_t)n * 512 + kk]) + h2d(wl[(size_t)n * 512 + kk]));
          acc_hi += att_hi[(size_t)i * D + kk] * h2d(wh[(size_t)n * 512 + kk]);
        }
        const size_t idx = ((size_t)b * S + i) * D + n;
        const double ref = (double)res[idx] + (double)bias[n] + acc * out_scale;
        const double ref_hi = (double)res[idx] + (double)bias[n] + acc_hi * out_scale;
        mag = std::max(mag, std::fabs(ref - (double)res[idx]));
        if (std::fabs((double)out[idx] - ref) > e_full) { err_out[3] = b; err_out[4] = i; err_out[5] = n; }
        e_full = std::max(e_full, std::fabs((double)out[idx] - ref));
        if ((zero_lo & 16) && std::fabs((double)out[idx] - ref) > 3e-6)
          printf("  row %d col %d (head %d d %d): got %.9g ref %.9g diff %.3e\n", i, n, n / dh, n % dh, (double)out[idx] - res[idx] - bias[n], ref - res[idx] - bias[n], (double)out[idx] - ref);
        err_out[6 + i / 32] = std::max(err_out[6 + i / 32], std::fabs((double)out[idx] - ref));          // per wave
        err_out[10 + n / 32] = std::max(err_out[10 + n / 32], std::fabs((double)out[idx] - ref));        // per column tile
        e_hi = std::max(e_hi, std::fabs(ref_hi - ref));
      }
  }
  err_out[0] = e_full / mag;
  err_out[1] = e_hi / mag;
  err_out[2] = mag;
  return 0;
}
