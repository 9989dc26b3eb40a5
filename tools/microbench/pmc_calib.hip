// Dev tool (MI355X): calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on KNOWN byte counts in the access patterns the
// sampling kernels use.  Every kernel moves each byte of a 1-GiB buffer (4x the 256-MiB Infinity Cache) exactly once.
//   hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o run -- ./pmc_calib     (then WRITE_SIZE)
// tools/pmc_calibrate.sh runs both passes and prints counter bytes / known bytes per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr size_t kBytes = 1ull << 30;

// 16 B per lane, lanes contiguous: the weight images' global side and every coalesced stream
__global__ void calib_read_x4(const float4* x, float* out, size_t n) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[threadIdx.x] = s;
}
// 4 B per lane, lanes contiguous
__global__ void calib_read_x1(const float* x, float* out, size_t n) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}
// global -> LDS DMA, 1 KiB per wave instruction: the weight stream of the stack kernel (ldm_dma.h dma_lin)
__global__ __launch_bounds__(256) void calib_read_lds_dma(const char* x, float* out, size_t n_kib) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned voff = (threadIdx.x & 63) * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"((unsigned)(wave * 1024)) : "memory");
  for (size_t k = (size_t)blockIdx.x * 4 + wave; k < n_kib; k += (size_t)gridDim.x * 4) {
    const char* g = x + k * 1024;
    asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(g) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (smem[threadIdx.x] == 77 && smem[threadIdx.x + 1] == 78) out[threadIdx.x] = 1.f;
}
// accumulator-layout row read: lane (r, hi) reads 16 B at columns 8 g + 4 hi of ITS OWN row of 464 floats (32 distinct
// rows per wave instruction) — the prologue of the non-loop stack kernel and the posterior kernel's embedding write
__global__ __launch_bounds__(256) void calib_read_acc_rows(const float* x, float* out, size_t n_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hi = lane >> 5;
  float s = 0.f;
  for (size_t row0 = ((size_t)blockIdx.x * 4 + wave) * 32; row0 + 32 <= n_rows; row0 += (size_t)gridDim.x * 128) {
    const float* p = x + (row0 + r) * 464 + hi * 4;
#pragma unroll 2
    for (int g = 0; g < 58; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(p + g * 8);
      s += v.x + v.y + v.z + v.w;
    }
  }
  if (s == 123.456f) out[threadIdx.x] = s;
}
__global__ void calib_write_x4(float4* x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void calib_write_x1(float* x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = 1.f;
}

int main() {
  char* buf;
  float* out;
  if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) return 1;
  (void)hipMemset(buf, 0, kBytes);
  (void)hipDeviceSynchronize();
  const size_t rows = kBytes / (464 * 4) / 128 * 128;
  hipLaunchKernelGGL(calib_read_x4, dim3(2048), dim3(256), 0, 0, (const float4*)buf, out, kBytes / 16);
  hipLaunchKernelGGL(calib_read_x1, dim3(2048), dim3(256), 0, 0, (const float*)buf, out, kBytes / 4);
  hipLaunchKernelGGL(calib_read_lds_dma, dim3(2048), dim3(256), 4096, 0, (const char*)buf, out, kBytes / 1024);
  hipLaunchKernelGGL(calib_read_acc_rows, dim3(2048), dim3(256), 0, 0, (const float*)buf, out, rows);
  hipLaunchKernelGGL(calib_write_x4, dim3(2048), dim3(256), 0, 0, (float4*)buf, kBytes / 16);
  hipLaunchKernelGGL(calib_write_x1, dim3(2048), dim3(256), 0, 0, (float*)buf, kBytes / 4);
  const hipError_t e = hipDeviceSynchronize();
  printf("known bytes: calib_read_x4 %zu calib_read_x1 %zu calib_read_lds_dma %zu calib_read_acc_rows %zu calib_write_x4 %zu calib_write_x1 %zu\n",
         kBytes, kBytes, kBytes, rows * 464 * 4, kBytes, kBytes);
  printf("status: %s\n", hipGetErrorString(e));
  return e == hipSuccess ? 0 : 1;
}
