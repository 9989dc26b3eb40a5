// Dev micro-benchmark (MI355X): what paces an LDS-fed MFMA stream with one wave per SIMD?
// One workgroup per CU (4 waves, 160 KiB of LDS requested => one wave per SIMD), every wave runs REP x 30 steps of
//     s_waitcnt lgkmcnt(PF-1) ; v_mfma_f32_32x32x16_f16 ; ds_read_b128 (item + PF)
// — the step of HeadStream / FfnStream (ldm_pipes.h) — in variants:
//   ACC  'v' accumulator in arch VGPRs | 'a' in AGPRs
//   BV   B operand changes every MFMA (29 register-resident fragments) | constant
//   NCH  accumulator chains (1, 2, 15 independent)
//   EXTRA n extra VALU (v_cvt_pk) per step, DMA 1: one global_load_lds_dwordx4 every 2nd step for the first 16 steps
//   hipcc --offload-arch=gfx950 -O3 -o mfma_feed mfma_feed.hip && ./mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int PF = 6, NIT = 30;

template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

template <bool AGPR, bool BVAR, int NCH, int EXTRA, bool DMA, int BAR = 0>
struct Pipe {
  f16x8 q[PF];
  f16x8 xf[30];
  f32x16 acc[NCH];
  unsigned aW[8];
  unsigned voff;
  const char* g;
  float junk;
  template <int IT>
  __device__ __forceinline__ void read_item() {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[IT % PF]) : "v"(aW[IT & 7]), "n"(256 * ((IT % NIT) >> 3)) : "memory");
  }
  template <int IT>
  __device__ __forceinline__ void step() {
    if constexpr (IT < NIT) {
      // LDS operations younger than item IT: PF - 1 reads + the epilogue stores issued in steps [IT - PF, IT - 1]
      constexpr int nw = (BAR & 4) ? ((IT >= 13 && IT <= 18) ? 2 : 0)
                       : (BAR & 8) ? ((IT >= 13 && IT <= 18) ? 1 : 0) + ((IT >= 14 && IT <= 19) ? 1 : 0) + ((IT >= 16 && IT <= 21) ? 1 : 0) + ((IT >= 17 && IT <= 22) ? 1 : 0)
                       : (BAR & 2) ? ((IT >= 13 && IT <= 18) ? 1 : 0) + ((IT >= 16 && IT <= 21) ? 1 : 0) : 0;
      wait_lgkm<PF - 1 + nw>();
      __builtin_amdgcn_sched_barrier(0);
      constexpr int c = IT % NCH;
      const f16x8 bop = BVAR ? xf[IT] : xf[0];
      if constexpr (NCH > 2) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[IT % PF], bop, acc[c], 0, 0, 0);
      else if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(q[IT % PF]), "v"(bop));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(q[IT % PF]), "v"(bop));
      __builtin_amdgcn_sched_barrier(0);
      read_item<IT + PF>();
      if constexpr (DMA && IT < 16) {
        if constexpr ((IT & 7) == 0) asm volatile("s_mov_b32 m0, %0" ::"s"(65536u + (IT >> 3) * 4096u) : "memory");
        else if constexpr ((IT & 1) == 1 && (IT & 7) < 8)
          asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(g), "n"(((IT & 7) >> 1) * 1024) : "memory");
      }
#pragma unroll
      for (int e = 0; e < EXTRA; ++e) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(junk));
      if constexpr ((BAR & 1) && IT == NIT - PF) {  // the streams' per-tile protocol
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if constexpr ((BAR & 2) && !(BAR & 12) && (IT == 12 || IT == 15)) {  // a tile epilogue's K / V^T store
        asm volatile("ds_write_b128 %0, %1" ::"v"(aW[IT & 7] + 98304u), "v"(xf[IT]) : "memory");
      }
      if constexpr ((BAR & 4) && IT == 12) {  // both stores in ONE MFMA gap
        asm volatile("ds_write_b128 %0, %1" ::"v"(aW[4] + 98304u), "v"(xf[12]) : "memory");
        asm volatile("ds_write_b128 %0, %1" ::"v"(aW[7] + 98304u), "v"(xf[15]) : "memory");
      }
      if constexpr ((BAR & 8) && (IT == 12 || IT == 13 || IT == 15 || IT == 16)) {  // the same bytes as four b64 stores
        typedef __attribute__((ext_vector_type(2))) float f2;
        const f2 half = __builtin_bit_cast(f2, __builtin_shufflevector(xf[IT], xf[IT], 0, 1, 2, 3));
        asm volatile("ds_write_b64 %0, %1" ::"v"(aW[IT & 7] + 98304u), "v"(half) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      step<IT + 1>();
    }
  }
};

template <bool AGPR, bool BVAR, int NCH, int EXTRA, bool DMA, int BAR = 0>
__global__ __launch_bounds__(256, 1) void bench(const char* g, float* out, unsigned long long* cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hi = lane >> 5;
  for (int i = threadIdx.x; i < 32768 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = reinterpret_cast<const float*>(g)[i];
  __syncthreads();
  Pipe<AGPR, BVAR, NCH, EXTRA, DMA, BAR> P;
  P.voff = lane * 16;
  P.g = g + blockIdx.x % 7 * 32768 + wave * 8192;
  P.junk = 1.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) P.aW[k] = r * 1024 + ((((k << 1) | hi) ^ (r & 15)) << 4);
#pragma unroll
  for (int i = 0; i < 30; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) P.xf[i][e] = (_Float16)(0.01f * ((lane + i * 3 + e) % 17) - 0.08f);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) P.acc[c][e] = 0.f;
  P.template read_item<0>(); P.template read_item<1>(); P.template read_item<2>();
  P.template read_item<3>(); P.template read_item<4>(); P.template read_item<5>();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < reps; ++it) P.template step<0>();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = P.junk;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += P.acc[c][e];
  if (s == 123.456f) out[threadIdx.x] = s;
  if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}

template <bool AGPR, bool BVAR, int NCH, int EXTRA, bool DMA, int BAR = 0>
void run(const char* name, const char* g, float* out, unsigned long long* cyc) {
  auto k = bench<AGPR, BVAR, NCH, EXTRA, DMA, BAR>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int reps = 200, blocks = 256;
  for (int pass = 0; pass < 2; ++pass) {
    hipMemset(cyc, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 160 * 1024, 0, g, out, cyc, reps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (pass) {
      const double per = (double)c / blocks / reps / NIT;
      printf("%-44s %6.1f cycles/MFMA   %.3f ms  => %.0f MHz-equivalent, %.0f TFLOP/s\n", name, per, ms,
             per * reps * NIT / (ms * 1e3), 4.0 * blocks * reps * NIT * 32768.0 / (ms * 1e-3) / 1e12);
    }
  }
}

int main() {
  char* g; float* out; unsigned long long* cyc;
  hipMalloc(&g, 8 * 32768); hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
  std::vector<_Float16> h(8 * 16384);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(0.02f * (float)((i * 2654435761u >> 20) % 23) - 0.2f);
  hipMemcpy(g, h.data(), 8 * 32768, hipMemcpyHostToDevice);
  run<false, true, 1, 0, false>("acc VGPR, B varies, 1 chain", g, out, cyc);
  run<true, true, 1, 0, false>("acc AGPR, B varies, 1 chain", g, out, cyc);
  run<false, false, 1, 0, false>("acc VGPR, B const, 1 chain", g, out, cyc);
  run<true, false, 1, 0, false>("acc AGPR, B const, 1 chain", g, out, cyc);
  run<false, true, 2, 0, false>("acc VGPR, B varies, 2 chains", g, out, cyc);
  run<true, true, 2, 0, false>("acc AGPR, B varies, 2 chains", g, out, cyc);
  run<true, false, 15, 0, false>("acc AGPR, B const, 15 chains", g, out, cyc);
  run<true, true, 15, 0, false>("acc AGPR, B varies, 15 chains", g, out, cyc);
  run<false, true, 1, 2, false>("acc VGPR, B varies, 1 chain, +2 VALU", g, out, cyc);
  run<true, true, 1, 2, false>("acc AGPR, B varies, 1 chain, +2 VALU", g, out, cyc);
  run<false, true, 1, 0, true>("acc VGPR, B varies, 1 chain, +DMA", g, out, cyc);
  run<true, true, 1, 0, true>("acc AGPR, B varies, 1 chain, +DMA", g, out, cyc);
  run<true, true, 2, 2, true>("acc AGPR, B varies, 2 chains, +2 VALU +DMA", g, out, cyc);
  run<false, true, 1, 0, false, 1>("acc VGPR, 1 chain, +barrier per 30", g, out, cyc);
  run<false, true, 1, 0, true, 1>("acc VGPR, 1 chain, +DMA +vmcnt(8) +barrier", g, out, cyc);
  run<false, true, 1, 0, true, 3>("acc VGPR, 1 chain, +DMA +barrier +2 ds_write", g, out, cyc);
  run<false, true, 1, 1, true, 3>("same +1 VALU per step (tile epilogue density)", g, out, cyc);
  run<false, true, 1, 0, true, 7>("+DMA +barrier, both ds_write_b128 in one gap", g, out, cyc);
  run<false, true, 1, 0, true, 11>("+DMA +barrier, 4 x ds_write_b64", g, out, cyc);
  run<false, true, 2, 0, true, 3>("2 chains, +DMA +barrier +2 ds_write_b128", g, out, cyc);
  return 0;
}
