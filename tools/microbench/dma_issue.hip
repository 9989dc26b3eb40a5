// Dev micro-benchmark (MI355X): what does ONE LDS-DMA piece cost the wave that issues it, by addressing form?
// The row-resident kernels (kernels_lngemm.hip) issue 16 pieces of 1 KiB per wave and 32-column tile between their MFMAs; the two-product build
// (8 pieces) showed each piece costs ~70 exposed cycles (profiles/r06_mixed_mode.txt part 5).  Is that the per-lane ADDRESS transfer?  Variants of
// the same stream — per tile 87 MFMAs on two chains, a piece behind every 5th MFMA, vmcnt(0) + barrier per tile, two 64-KiB stages:
//   0  no DMA
//   1  global_load_lds_dwordx4 v_off, s[base:base+1] offset:imm          (the shipping form: a 32-bit offset VGPR per lane)
//   2  buffer_load_dwordx4 off, s[srd:srd+3], s_off offset:imm lds        (ADD_TID_ENABLE, stride 16: NO address VGPR — lane i reads base + 16 i)
//   3  buffer_load_dwordx4 v_off, s[srd:srd+3], s_off offen offset:imm lds (buffer form WITH the offset VGPR: separates "buffer" from "no VGPR")
// Variant 2's LDS image is checked against the source.   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_issue dma_issue.hip && /tmp/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int STAGE = 65536, NMF = 87, NP = 16;

template <int V, int J>
__device__ __forceinline__ void piece(unsigned voff, const char* g, u32x4 srd, unsigned soff, unsigned m0) {
  if constexpr (V == 0) return;
  if constexpr ((J & 3) == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0 + (J >> 2) * 4096) : "memory");
  if constexpr (V == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(g + (J >> 2) * 4096), "n"((J & 3) * 1024) : "memory");
  if constexpr (V == 2) asm volatile("buffer_load_dwordx4 off, %0, %1 offset:%2 lds" ::"s"(srd), "s"(soff + (J >> 2) * 4096), "n"((J & 3) * 1024) : "memory");
  if constexpr (V == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(voff), "s"(srd), "s"(soff + (J >> 2) * 4096), "n"((J & 3) * 1024) : "memory");
}

template <int V, int I>
__device__ __forceinline__ void mfmas(f32x16& a, f32x16& b, f16x8 x, f16x8 w, unsigned voff, const char* g, u32x4 srd, unsigned soff, unsigned m0) {
  if constexpr (I < NMF) {
    if constexpr (I & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(b) : "v"(w), "v"(x));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a) : "v"(w), "v"(x));
    if constexpr (I % 5 == 1 && I / 5 < NP) piece<V, I / 5>(voff, g, srd, soff, m0);
    mfmas<V, I + 1>(a, b, x, w, voff, g, srd, soff, m0);
  }
}

template <int V>
__global__ __launch_bounds__(256, 1) void dma_issue(const char* img, int n_tiles, int img_tiles, unsigned long long* cyc, float* sink, char* lds_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  f32x16 a, b;
  for (int k = 0; k < 16; ++k) a[k] = b[k] = 0.f;
  f16x8 x, w;
  for (int k = 0; k < 8; ++k) { x[k] = (_Float16)(0.001f * (lane + k)); w[k] = (_Float16)(0.002f * (lane - k)); }
  const unsigned voff = lane * 16;
  const char* base = img + wave * 16384;
  u32x4 srd;
  {
    const unsigned long long p = (unsigned long long)base;
    srd.x = (unsigned)p;
    srd.y = (unsigned)(p >> 32) | (V == 2 ? (16u << 16) : 0u);   // stride 16 (ADD_TID: lane i at + 16 i)
    srd.z = 0xffffffffu;                                         // num_records
    srd.w = (V == 2 ? (1u << 23) : 0u);                          // ADD_TID_ENABLE; DATA_FORMAT 0 (= stride[17:14] in this mode)
  }
  asm volatile("" : "+s"(srd));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < n_tiles; ++t) {
    const int ti = t % img_tiles;
    const unsigned soff = (unsigned)ti * STAGE;
    const unsigned m0 = lds0 + (t & 1) * STAGE + wave * 16384;
    mfmas<V, 0>(a, b, x, w, voff, base + (size_t)ti * STAGE, srd, soff, m0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a), "+v"(b));
  if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
  float s = 0.f;
  for (int k = 0; k < 16; ++k) s += a[k] + b[k];
  if (s == 123.456f) sink[0] = s;
  if (lds_out && blockIdx.x == 0) {   // the stage the last tile was loaded into
    __syncthreads();
    const int st = (n_tiles - 1) & 1;
    for (int i = threadIdx.x; i < STAGE / 16; i += 256) reinterpret_cast<float4*>(lds_out)[i] = reinterpret_cast<const float4*>(smem + st * STAGE)[i];
  }
}

template <int V>
void run(const char* name, const char* d_img, int img_tiles, const std::vector<char>& h_img) {
  int n_cu = 0;
  hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
  unsigned long long* cyc;
  float* sink;
  char* lds_out;
  hipMalloc(&cyc, 8); hipMalloc(&sink, 4); hipMalloc(&lds_out, STAGE);
  hipFuncSetAttribute((const void*)dma_issue<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  const int n_tiles = 58;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(cyc, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(dma_issue<V>, dim3(n_cu), dim3(256), 2 * STAGE, 0, d_img, n_tiles, img_tiles, cyc, sink, lds_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (rep == 2) {
      printf("%-58s %7.1f us per launch of %d tiles = %6.3f us per tile; s_memtime ticks per tile and workgroup %.1f", name, ms * 1e3, n_tiles, ms * 1e3 / n_tiles,
             (double)c / n_cu / n_tiles);
      if (V != 0) {
        std::vector<char> got(STAGE);
        hipMemcpy(got.data(), lds_out, STAGE, hipMemcpyDeviceToHost);
        const int ti = (n_tiles - 1) % img_tiles;
        printf("   LDS image == source: %s", memcmp(got.data(), h_img.data() + (size_t)ti * STAGE, STAGE) == 0 ? "yes" : "NO");
      }
      printf("\n");
    }
  }
  hipFree(cyc); hipFree(sink); hipFree(lds_out);
}

int main() {
  const int img_tiles = 58;
  std::vector<char> h((size_t)img_tiles * STAGE);
  unsigned s = 12345;
  for (auto& c : h) { s = s * 1664525u + 1013904223u; c = (char)(s >> 24); }
  char* d;
  hipMalloc(&d, h.size());
  hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  run<0>("0 no DMA", d, img_tiles, h);
  run<1>("1 global_load_lds_dwordx4 v_off, s[base]", d, img_tiles, h);
  run<2>("2 buffer_load_dwordx4 off, srd, s_off lds (ADD_TID, no VGPR)", d, img_tiles, h);
  run<3>("3 buffer_load_dwordx4 v_off, srd, s_off offen lds", d, img_tiles, h);
  return 0;
}
