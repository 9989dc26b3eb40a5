// Dev micro-benchmark (MI355X): what does the chip sustain for the loop kernel's INSTRUCTION MIX and PHASE STRUCTURE?
// ("a mfma_feed variant with the kernel's exact VALU : MFMA : LDS ratio that reproduces 0.42 and so
// proves the ceiling — or shows the headroom".)
//
// stack_stream_k<false,2> per launch (profiles/r03_final2_sq_counters_fast_loop.txt): 4.62e9 MFMA, 1.41e10 VALU (3.06 per
// MFMA), 5.37e9 LDS (1.16 per MFMA), matrix pipes busy 0.62 of the cycles at ~1.69 GHz.  Its phases are of two kinds: MFMA
// streams (one v_mfma_f32_32x32x16_f16 per ~33 cycles with its ds_read_b128 and a few VALU in the shadow) and VALU-only
// phases (row statistics, LayerNorm transforms, softmax, the step's tail) during which the matrix pipe idles.
//
// One workgroup per CU, one wave per SIMD (160 KiB of LDS requested), every wave repeats
//     burst:  NB x [ s_waitcnt lgkmcnt ; v_mfma ; ds_read_b128 ; E x VALU in the MFMA's shadow ]
//     gap:    G x VALU with the matrix pipe idle (v_fma_f32 / v_pk_mul_f32 / v_exp_f32 in the kernel's proportions)
// for a grid of (E, G).  Printed: cycles per MFMA, matrix-pipe duty (33 cycles per MFMA / total), effective clock (cycles
// / wall time), TFLOP/s and fraction of the 2.5 PFLOP/s peak.  The kernel's point is E ~ 1.6, duty 0.62.
//   hipcc --offload-arch=gfx950 -O3 -o mix_feed mix_feed.hip && ./mix_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int PF = 6, NB = 30;

template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

template <int E, int G>
__global__ __launch_bounds__(256, 1) void bench(const char* g, float* out, unsigned long long* cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
  for (int i = threadIdx.x; i < 32768 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = reinterpret_cast<const float*>(g)[i];
  __syncthreads();
  f16x8 q[PF], xf[NB];
  f32x16 acc0, acc1;
  unsigned aW[8];
  float v0 = 1.0f + lane * 1e-3f, v1 = 0.5f, v2 = 0.25f, v3 = 2.0f;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 p0 = {1.0f, 2.0f}, p1 = {0.5f, 0.25f};
#pragma unroll
  for (int k = 0; k < 8; ++k) aW[k] = r * 1024 + ((((k << 1) | hi) ^ (r & 15)) << 4);
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) xf[i][e] = (_Float16)(0.01f * ((lane + i * 3 + e) % 17) - 0.08f);
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
  for (int i = 0; i < PF; ++i)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(aW[i & 7]), "n"(256 * (i >> 3)) : "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      wait_lgkm<PF - 1>();
      __builtin_amdgcn_sched_barrier(0);
      if (i & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc1) : "v"(q[i % PF]), "v"(xf[i]));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc0) : "v"(q[i % PF]), "v"(xf[i]));
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i % PF]) : "v"(aW[(i + PF) & 7]), "n"(256 * (((i + PF) % NB) >> 3)) : "memory");
#pragma unroll
      for (int e = 0; e < E; ++e) {  // shadow VALU: the streams' casts / packed scale-and-shift
        if (e & 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(p1));
        else asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v2) : "v"(v3));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // VALU-only phase (matrix pipe idle): fma chains on 4 independent registers, a packed op and a transcendental per 8
#pragma unroll 8
    for (int k = 0; k < G / 8; ++k) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v3));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(v1), "v"(v3));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(p1));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v3));
      asm volatile("v_max_f32 %0, %0, %1" : "+v"(v2) : "v"(v1));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(p1));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v3));
      asm volatile("v_exp_f32 %0, %1" : "=v"(v3) : "v"(v1));
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = v0 + v2 + v3 + p0[0] + p0[1];
#pragma unroll
  for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
  if (s == 123.456f) out[threadIdx.x] = s;
  if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}

template <int E, int G>
void run(const char* g, float* out, unsigned long long* cyc) {
  auto k = bench<E, G>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int reps = 400, blocks = 256;
  for (int pass = 0; pass < 2; ++pass) {
    hipMemset(cyc, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int l = 0; l < 8; ++l) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 160 * 1024, 0, g, out, cyc, reps);  // ~40 ms: DVFS settles
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (pass) {
      const double per_rep = (double)c / blocks / reps / 8;                 // s_memtime ticks (100 MHz) per rep ...
      const double wall_rep_ns = ms * 1e6 / reps / 8;                       // ... and wall time per rep
      const double tf = 4.0 * blocks * reps * 8 * NB * 32768.0 / (ms * 1e-3) / 1e12;
      (void)per_rep;
      printf("E=%d shadow VALU/MFMA  G=%4d gap VALU  (%.2f VALU/MFMA total): %7.1f ns per %d-MFMA burst+gap => %6.0f TFLOP/s = %.3f of peak\n",
             E, G, E + (double)G / NB, wall_rep_ns, NB, tf, tf / 2500.0);
    }
  }
}

int main() {
  char* g; float* out; unsigned long long* cyc;
  hipMalloc(&g, 8 * 32768); hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
  std::vector<_Float16> h(8 * 16384);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(0.02f * (float)((i * 2654435761u >> 20) % 23) - 0.2f);
  hipMemcpy(g, h.data(), 8 * 32768, hipMemcpyHostToDevice);
  printf("# burst = %d MFMA (32x32x16 f16) each with one ds_read_b128; one wave per SIMD on all 256 CUs\n", NB);
  run<0, 0>(g, out, cyc);
  run<2, 0>(g, out, cyc);
  run<3, 0>(g, out, cyc);
  run<0, 48>(g, out, cyc);
  run<2, 48>(g, out, cyc);
  run<0, 96>(g, out, cyc);
  run<2, 32>(g, out, cyc);
  run<2, 64>(g, out, cyc);
  run<2, 96>(g, out, cyc);
  run<2, 160>(g, out, cyc);
  run<1, 64>(g, out, cyc);
  run<3, 64>(g, out, cyc);
  return 0;
}
