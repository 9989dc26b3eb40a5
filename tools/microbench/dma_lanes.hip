// Dev micro-benchmark (MI355X): is the per-CU LDS-DMA fill limit (37 - 40 GB/s) a limit in BYTES / LANES or in
// wave-instructions?  Every CU streams a 23-MiB image (the weight images of one reverse step) into a 2 x 64 KiB LDS ring with
// global_load_lds_dwordx4 pieces (1 KiB per wave-instruction when all 64 lanes are active), one workgroup of 4 waves per CU as in the
// stack / lngemm kernels, 16 pieces per wave and stage, vmcnt(0) + barrier per stage — with EXEC restricted to the first NL lanes of
// every piece (64, 58 = the 464 real k of a 512-wide tile row, 48, 32).  If the time follows the active lanes, the pad lanes of the tile
// rows (9.4 % of every piece of the in_proj / W1 tiles) are worth masking.
//   hipcc --offload-arch=gfx950 -O3 -o dma_lanes dma_lanes.hip && ./dma_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int STAGE = 65536;

__global__ __launch_bounds__(256, 1) void fill_k(const char* img, int n_stages, int reps, unsigned long long mask, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned voff = lane * 16;
  for (int r = 0; r < reps; ++r)
    for (int st = 0; st < n_stages; ++st) {
      const char* g = img + (size_t)st * STAGE + wave * 16384;
      const unsigned l = lds0 + (st & 1) * STAGE + wave * 16384;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        asm volatile(
            "s_mov_b32 m0, %2\n\ts_nop 0\n\ts_mov_b64 exec, %3\n\t"
            "global_load_lds_dwordx4 %0, %1\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:3072\n\t"
            "s_mov_b64 exec, -1" ::"v"(voff),
            "s"(g + q * 4096), "s"(l + q * 4096), "s"(mask)
            : "memory");
      }
      if (st & 1) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the previous stage of this wave has landed
        __builtin_amdgcn_s_barrier();
      }
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem + 4 * (blockIdx.x & 1023));
}

int main() {
  const int n_stages = 368;   // 23 MiB
  char* img;
  unsigned* sink;
  hipMalloc(&img, (size_t)n_stages * STAGE);
  hipMemset(img, 1, (size_t)n_stages * STAGE);
  hipMalloc(&sink, 1024 * 4);
  hipFuncSetAttribute((const void*)fill_k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int lanes[] = {64, 58, 48, 32, 64};
  for (int grid : {256, 128})
    for (int nl : lanes) {
      const unsigned long long mask = nl == 64 ? ~0ull : ((1ull << nl) - 1);
      const int reps = 4;
      hipLaunchKernelGGL(fill_k, dim3(grid), dim3(256), 2 * STAGE, 0, img, n_stages, 1, mask, sink);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(fill_k, dim3(grid), dim3(256), 2 * STAGE, 0, img, n_stages, reps, mask, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double pieces = (double)reps * n_stages * 64;                 // per workgroup
      const double bytes = pieces * nl * 16;                              // useful bytes per workgroup
      printf("workgroups %3d  active lanes %2d: %8.3f ms  %6.1f ns per piece and CU  %6.2f GB/s per CU (active bytes)  %6.2f TB/s chip\n", grid, nl, ms,
             ms * 1e6 / pieces, bytes / (ms * 1e-3) / 1e9, bytes * grid / (ms * 1e-3) / 1e12);
    }
  return 0;
}
