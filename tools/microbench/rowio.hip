// Dev micro-benchmark (MI355X): how fast can one workgroup per layout (4 waves x 32 rows x 464 fp32) read / write its
// rows in (a) the MFMA accumulator layout the fused layer kernel uses directly (a lane touches 16 B of its own row
// per instruction: 32 distinct cache lines, 32 B each, per wave instruction) versus (b) a line-coalesced pattern
// (8 rows x 128 B per wave instruction) with a wave-local LDS transpose to the accumulator layout.
//   hipcc --offload-arch=gfx950 -O3 -o rowio rowio.hip && ./rowio
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int S = 125, N = 464, NG = 58;

__global__ __launch_bounds__(256, 1) void read_acc(const float* x, float* out) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hi = lane >> 5;
  const int row = wave * 32 + r;
  const size_t m = (size_t)blockIdx.x * S + (row < S ? row : S - 1);
  const float* p = x + m * N + hi * 4;
  float4 v[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) v[g] = *reinterpret_cast<const float4*>(p + g * 8);
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) s += v[g].x + v[g].y + v[g].z + v[g].w;
  if (s == 123.456f) out[threadIdx.x] = s;
}

// line-coalesced read: instruction i of tile t: lane L -> row 8i + L/8, 16-B chunk L%8 of the tile's 128 B
__global__ __launch_bounds__(256, 1) void read_line(const float* x, float* out, int transpose) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hi = lane >> 5;
  char* stg = smem + wave * (32 * 144 * 4);  // 4 tiles in flight, rows padded to 144 B
  const int lr = lane >> 3, ch = lane & 7;
  float s = 0.f;
  for (int t0 = 0; t0 < 15; t0 += 4) {
    float4 v[4][4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int t = t0 + tt;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + 8 * i + lr;
        const size_t m = (size_t)blockIdx.x * S + (row < S ? row : S - 1);
        int col = t * 32 + ch * 4;
        if (t >= 15 || col >= N) col = N - 4;
        v[tt][i] = *reinterpret_cast<const float4*>(x + m * N + col);
      }
    }
    if (transpose) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float4*>(stg + tt * (32 * 144) + (8 * i + lr) * 144 + ch * 16) = v[tt][i];
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 a = *reinterpret_cast<const float4*>(stg + tt * (32 * 144) + r * 144 + (2 * g + hi) * 16);
          s += a.x + a.y + a.z + a.w;
        }
    } else {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += v[tt][i].x + v[tt][i].y + v[tt][i].z + v[tt][i].w;
    }
  }
  if (s == 123.456f) out[threadIdx.x] = s;
}

__global__ __launch_bounds__(256, 1) void write_acc(float* x, float seed) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hi = lane >> 5;
  const int row = wave * 32 + r;
  const bool valid = row < S;
  const size_t m = (size_t)blockIdx.x * S + (valid ? row : S - 1);
  float* p = x + m * N + hi * 4;
#pragma unroll
  for (int g = 0; g < NG; ++g)
    if (valid) *reinterpret_cast<float4*>(p + g * 8) = make_float4(seed + g, seed, seed + lane, seed);
}

__global__ __launch_bounds__(256, 1) void write_line(float* x, float seed) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hi = lane >> 5;
  char* stg = smem + wave * (32 * 144 * 4);
  const int lr = lane >> 3, ch = lane & 7;
  for (int t0 = 0; t0 < 15; t0 += 4) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(stg + tt * (32 * 144) + r * 144 + (2 * g + hi) * 16) =
            make_float4(seed + g, seed + tt, seed + lane, seed + t0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int t = t0 + tt;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(stg + tt * (32 * 144) + (8 * i + lr) * 144 + ch * 16);
        const int row = wave * 32 + 8 * i + lr;
        const int col = t * 32 + ch * 4;
        if (t < 15 && col < N && row < S)
          *reinterpret_cast<float4*>(x + ((size_t)blockIdx.x * S + row) * N + col) = a;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
}

template <class F>
static float time_it(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / reps;
}

int main() {
  const int B = 512;
  float *x, *out;
  // 4 distinct buffers cycled so that reads are not served from a warm L2 of the same launch's predecessor
  const size_t elems = (size_t)B * S * N;
  hipMalloc(&x, elems * 4 * 4);
  hipMalloc(&out, 4096);
  hipMemset(x, 0, elems * 4 * 4);
  const size_t lds = 150 * 1024;
  hipFuncSetAttribute((const void*)read_acc, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)read_line, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)write_acc, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)write_line, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int blocks : {64, 128, 256, 512}) {
    int k = 0;
    auto buf = [&]() { return x + (size_t)((k++) & 3) * elems; };
    const double mb = (double)blocks * S * N * 4 / 1e6;
    float t;
    t = time_it([&] { hipLaunchKernelGGL(read_acc, dim3(blocks), dim3(256), lds, 0, buf(), out); }, 40);
    printf("blocks %3d  read  acc-layout        %7.2f us  %6.0f GB/s\n", blocks, t, mb / t);
    t = time_it([&] { hipLaunchKernelGGL(read_line, dim3(blocks), dim3(256), lds, 0, buf(), out, 0); }, 40);
    printf("blocks %3d  read  line (no xpose)   %7.2f us  %6.0f GB/s\n", blocks, t, mb / t);
    t = time_it([&] { hipLaunchKernelGGL(read_line, dim3(blocks), dim3(256), lds, 0, buf(), out, 1); }, 40);
    printf("blocks %3d  read  line + LDS xpose  %7.2f us  %6.0f GB/s\n", blocks, t, mb / t);
    t = time_it([&] { hipLaunchKernelGGL(write_acc, dim3(blocks), dim3(256), lds, 0, buf(), 1.0f); }, 40);
    printf("blocks %3d  write acc-layout        %7.2f us  %6.0f GB/s\n", blocks, t, mb / t);
    t = time_it([&] { hipLaunchKernelGGL(write_line, dim3(blocks), dim3(256), lds, 0, buf(), 1.0f); }, 40);
    printf("blocks %3d  write LDS xpose + line  %7.2f us  %6.0f GB/s\n", blocks, t, mb / t);
  }
  hipError_t e = hipDeviceSynchronize();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
