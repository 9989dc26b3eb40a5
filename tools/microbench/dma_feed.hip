// Dev micro-benchmark (MI355X): what does the weight LDS-DMA cost an LDS-fed MFMA stream?
// The FFN chunk stream of the stack kernel (ldm_pipes.h FfnStream) runs its GEMM1 (29 MFMAs, B operand changes, two
// VGPR accumulator chains, one 1-KiB global_load_lds_dwordx4 every 2nd step) at 44 cycles per MFMA and its GEMM2 (30
// MFMAs, 15 AGPR accumulators, no DMA) at 32 (profiles/r02_call8_*, r02_call33_stack_phase_probe.txt).  This bench runs the
// same 60-step chunk (59 MFMAs + the ReLU pseudo step, 6-deep ds_read_b128 queue, vmcnt(0) + barrier at step 54, two
// 64-KiB stages) on every CU, one wave per SIMD, and varies ONLY where the 16 DMA pieces per wave and chunk are issued
// and where their bytes come from:
//   PLACE 0 none | 1 odd steps 1..31 (the shipping schedule) | 2 every 3rd step 2..47 | 3 odd steps 29..59 (inside GEMM2)
//         4 pairs: two pieces every 4th step 3..31 | 5 odd steps 1..31 but only 8 pieces (half the bytes)
//   SRC   0 the same 64 KiB every chunk (L1/L2 hot) | 1 a 23-MiB image walked chunk by chunk by every workgroup (the weights)
//   hipcc --offload-arch=gfx950 -O3 -o dma_feed dma_feed.hip && ./dma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int PF = 6, NIT = 60, KS = 29, NT2 = 15, SYNC = NIT - PF;
constexpr int STAGE = 65536;

template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

template <int PLACE>
__host__ __device__ constexpr int piece_at(int it) {  // index of the piece issued at step it, or -1
  if (PLACE == 1) return (it % 2 == 1 && it / 2 < 16) ? it / 2 : -1;
  if (PLACE == 2) return (it % 3 == 2 && it / 3 < 16) ? it / 3 : -1;
  if (PLACE == 3) return (it >= 29 && it % 2 == 1 && (it - 29) / 2 < 16) ? (it - 29) / 2 : -1;
  if (PLACE == 5) return (it % 2 == 1 && it / 2 < 16 && (it / 2) % 2 == 0) ? it / 2 : -1;
  return -1;
}

template <int PLACE, bool PIPE = false>
struct Pipe {
  f16x8 q[PF];
  f16x8 xf[KS];
  f32x16 ha, hb;
  f32x16 acc[NT2];
  f16x8 pf[2], pfn[2];
  unsigned aW1[8], aW2[2];
  unsigned voff, mnext;
  const char* gnext;

  template <int IT>
  __device__ __forceinline__ void read_item() {
    constexpr int I = IT % NIT;
    if constexpr (I <= KS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[IT % PF]) : "v"(aW1[I & 7]), "n"(256 * ((I % KS) >> 3)) : "memory");
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[IT % PF]) : "v"(aW2[(I - KS - 1) / NT2]), "n"(32768 + ((I - KS - 1) % NT2) * 2048) : "memory");
  }
  template <int P>
  __device__ __forceinline__ void piece() {  // piece P of this wave's 16 KiB of the next stage
    if constexpr ((P & 3) == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(mnext + (P >> 2) * 4096) : "memory");
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(gnext + (P >> 2) * 4096), "n"((P & 3) * 1024) : "memory");
  }
  template <int IT>
  __device__ __forceinline__ void step() {
    if constexpr (IT < NIT) {
      wait_lgkm<PF - 1>();
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 cur = q[IT % PF];
      if constexpr (IT < KS) {
        if constexpr (IT % 2 == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ha) : "v"(cur), "v"(xf[IT]));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(hb) : "v"(cur), "v"(xf[IT]));
        if constexpr (IT == KS - 1 && !PIPE) asm volatile("s_nop 15" ::: "memory");
        if constexpr (IT == 0 && PIPE) { pf[0] = pfn[0]; pf[1] = pfn[1]; }  // the fragments finished during the last GEMM2
      } else if constexpr (IT == KS) {
        if constexpr (!PIPE) {
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
#pragma unroll
            for (int e = 0; e < 4; ++e) pf[rq >> 1][(rq & 1) * 4 + e] = (_Float16)fmaxf(ha[rq * 4 + e] + hb[rq * 4 + e], 0.f);
        }
      } else {
        constexpr int sx = (IT - KS - 1) / NT2, t = (IT - KS - 1) % NT2;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, pf[sx], acc[t], 0, 0, 0);
        if constexpr (PIPE) {
          // software-pipelined FFN: this GEMM2 multiplies the PREVIOUS chunk's hidden fragments; the ReLU / cast of the
          // GEMM1 that just finished runs in the shadows of GEMM2 steps 2..9 (one accumulator pair each)
          constexpr int g = IT - KS - 1;
          if constexpr (g >= 2 && g < 10) {
            constexpr int i = g - 2;
            typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v;
            const float s0 = ha[2 * i] + hb[2 * i], s1 = ha[2 * i + 1] + hb[2 * i + 1];
            f16x2v hv = {(_Float16)s0, (_Float16)s1};
            const f16x2v zero = {(_Float16)0.f, (_Float16)0.f};
            hv = __builtin_elementwise_max(hv, zero);
            pfn[i >> 2][(i & 3) * 2 + 0] = hv[0];
            pfn[i >> 2][(i & 3) * 2 + 1] = hv[1];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (IT == SYNC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) aW2[sx] ^= 0x10000u;
      }
      read_item<IT + PF>();
      if constexpr (IT == KS - PF) {
#pragma unroll
        for (int k = 0; k < 8; ++k) aW1[k] ^= 0x10000u;
      }
      if constexpr (PLACE == 4) {
        if constexpr (IT % 4 == 3 && IT / 4 < 8) {
          piece<2 * (IT / 4)>();
          piece<2 * (IT / 4) + 1>();
        }
      } else if constexpr (piece_at<PLACE>(IT) >= 0) {
        piece<piece_at<PLACE>(IT)>();
      }
      __builtin_amdgcn_sched_barrier(0);
      step<IT + 1>();
    }
  }
};

template <int PLACE, int SRC, bool PIPE = false>
__global__ __launch_bounds__(256, 1) void bench(const char* g, int n_img_chunks, float* out, unsigned long long* cyc, int chunks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hi = lane >> 5;
  for (int i = threadIdx.x; i < 2 * STAGE / 4; i += 256) reinterpret_cast<float*>(smem)[i] = reinterpret_cast<const float*>(g)[i];
  __syncthreads();
  Pipe<PLACE, PIPE> P;
#pragma unroll
  for (int e = 0; e < 8; ++e) P.pfn[0][e] = P.pfn[1][e] = P.pf[0][e] = P.pf[1][e] = (_Float16)0.25f;
  P.voff = lane * 16;
#pragma unroll
  for (int k = 0; k < 8; ++k) P.aW1[k] = r * 1024 + ((((k << 1) | hi) ^ (r & 15)) << 4);
#pragma unroll
  for (int sx = 0; sx < 2; ++sx) P.aW2[sx] = r * 64 + (((2 * sx + hi) ^ ((r >> 2) & 3)) << 4);
#pragma unroll
  for (int i = 0; i < KS; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) P.xf[i][e] = (_Float16)(0.01f * ((lane + i * 3 + e) % 17) - 0.08f);
#pragma unroll
  for (int c = 0; c < NT2; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) P.acc[c][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) P.ha[e] = P.hb[e] = 0.f;
  P.template read_item<0>(); P.template read_item<1>(); P.template read_item<2>();
  P.template read_item<3>(); P.template read_item<4>(); P.template read_item<5>();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int c = 0; c < chunks; ++c) {
    const int src_chunk = SRC ? (c + 1) % n_img_chunks : 0;
    P.gnext = g + (size_t)src_chunk * STAGE + wave * 16384;
    P.mnext = ((c + 1) & 1) * STAGE + wave * 16384;
    P.template step<0>();
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NT2; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += P.acc[c][e];
  if (s == 123.456f) out[threadIdx.x] = s;
  if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}

template <int PLACE, int SRC, bool PIPE = false>
void run(const char* name, const char* g, int n_img_chunks, float* out, unsigned long long* cyc) {
  auto k = bench<PLACE, SRC, PIPE>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int chunks = 232 * 4, blocks = 256;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 156 * 1024, 0, g, n_img_chunks, out, cyc, chunks);
  hipDeviceSynchronize();
  hipMemset(cyc, 0, 8);
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 156 * 1024, 0, g, n_img_chunks, out, cyc, chunks);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per_chunk = (double)c / blocks / chunks;
  const double tf = 2.0 * 32 * 32 * 16 * 59 * 4 * (double)blocks * chunks / (ms * 1e-3) / 1e12;
  printf("%-78s %7.1f cycles/chunk  %5.1f cycles/MFMA  %7.3f ms  clock %4.0f MHz  %6.0f TFLOP/s\n", name, per_chunk, per_chunk / 59.0, ms,
         per_chunk * chunks / (ms * 1e3), tf);
}

int main() {
  const int n_img_chunks = 360;  // 360 x 64 KiB = 23.6 MB, the fast mode's weight images of one reverse step
  char* g;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&g, (size_t)n_img_chunks * STAGE);
  hipMalloc(&out, 1024);
  hipMalloc(&cyc, 8);
  std::vector<unsigned short> h((size_t)n_img_chunks * STAGE / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x2000 + (unsigned short)((i * 2654435761u) >> 22);  // small fp16 values
  hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  run<0, 0>("no DMA", g, n_img_chunks, out, cyc);
  run<1, 0>("16 pieces at odd steps 1..31 (shipping), hot 64 KiB source", g, n_img_chunks, out, cyc);
  run<1, 1>("16 pieces at odd steps 1..31 (shipping), 23.6 MB image walked by every workgroup", g, n_img_chunks, out, cyc);
  run<2, 1>("16 pieces at every 3rd step 2..47, 23.6 MB image", g, n_img_chunks, out, cyc);
  run<3, 1>("16 pieces at odd steps 29..59 (inside GEMM2), 23.6 MB image", g, n_img_chunks, out, cyc);
  run<4, 1>("8 pairs of pieces at steps 3, 7, .. 31, 23.6 MB image", g, n_img_chunks, out, cyc);
  run<5, 1>("8 pieces at steps 1, 5, .. 29 (half the bytes), 23.6 MB image", g, n_img_chunks, out, cyc);
  run<0, 0, true>("PIPELINED (ReLU inside the next GEMM2), no DMA", g, n_img_chunks, out, cyc);
  run<1, 1, true>("PIPELINED, 16 pieces at odd steps 1..31, 23.6 MB image", g, n_img_chunks, out, cyc);
  run<1, 1>("16 pieces at odd steps 1..31 (shipping), 23.6 MB image (again)", g, n_img_chunks, out, cyc);
  run<0, 0>("no DMA (again)", g, n_img_chunks, out, cyc);
  return 0;
}
