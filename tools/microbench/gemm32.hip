// Dev tool (MI355X): where the time of the exact mode's fp32 GEMM goes, and what a persistent / phase-staggered schedule
// of the same tile returns.  M = 32 000 rows (a chunk of 256 layouts), the four shapes of a layer.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm32 gemm32.hip && ./gemm32
// Variants of ONE 128 x 128 tile kernel (the shipping gemm_f32_128x128: 4 waves, 2 x 2 MFMA 32x32x2 tiles each, BK = 16,
// double-buffered LDS, 4 workgroups per CU):
//   base        one workgroup per tile, grid = tiles (the shipping schedule)
//   pers        1 024 resident workgroups, each walks tiles w, w + G, ...
//   pers+stag   the same, workgroup slot s of a CU (= blockIdx / 256) starts s quarter-tiles late: the four workgroups of
//               a CU reach their epilogues at different times
//   noepi       base without the epilogue's memory traffic (ablation)      k1: base with ONE k-tile (prologue + epilogue)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BK = 16, LD = BK + 4;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

struct Epi {
  const float* bias;
  const float* res;
  float* C;
  int M, N, ldres, ldc, relu;
};

// MODE bit 0: persistent; bit 1: stagger; bit 2: no epilogue traffic; main-loop ablations: bit 3: no global -> LDS staging
// inside the loop; bit 4: no barrier inside the loop; bit 5: no LDS fragment reads (operands = whatever the registers hold);
// bit 6: global loads but no LDS writes inside the loop; bit 7: LDS writes (of stale registers) but no global loads;
// bit 9: the LDS writes of the next k-tile sit between the two MFMA groups of the iteration instead of behind them
// bit 8: the A rows wrap at 256 (the operand stream is L2-resident: what remains is the cost of the instructions)
template <int MODE>
__global__ __launch_bounds__(256, 4) void g32(const float* __restrict__ A, const float* __restrict__ W, int lda, int ldw,
                                              int K, int tiles_n, int n_tiles, int stagger_cycles, Epi e) {
  __shared__ __attribute__((aligned(16))) float As[2][128][LD];
  __shared__ __attribute__((aligned(16))) float Ws[2][128][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 2, lc4 = (tid & 3) * 4;
  const int frow = lane & 31, fk = (lane >> 5) * 4;
  const int nk = K / BK;
  if (MODE & 2) {
    const int slot = blockIdx.x / 256;  // (dispatch order: the first 256 workgroups take one slot of every CU)
    for (long long t0 = clock64(); clock64() - t0 < (long long)slot * stagger_cycles;) __builtin_amdgcn_s_sleep(32);
  }
  for (int t = blockIdx.x; t < n_tiles; t += (MODE & 1) ? gridDim.x : n_tiles) {
    const int tile = xcd_remap(t, n_tiles);
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
    float4 ra[2], rw[2];
    auto gload = [&](int kt) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + lrow + i * 64, n = n0 + lrow + i * 64;
        const int ma = (MODE & 256) ? (m & 255) : m;
        ra[i] = (m < e.M) ? *reinterpret_cast<const float4*>(A + (size_t)ma * lda + kt * BK + lc4) : make_float4(0, 0, 0, 0);
        rw[i] = (n < e.N) ? *reinterpret_cast<const float4*>(W + (size_t)n * ldw + kt * BK + lc4) : make_float4(0, 0, 0, 0);
      }
    };
    auto lstore = [&](int buf) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<float4*>(&As[buf][lrow + i * 64][lc4]) = ra[i];
        *reinterpret_cast<float4*>(&Ws[buf][lrow + i * 64][lc4]) = rw[i];
      }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    gload(0);
    if (MODE & 1) __syncthreads();  // (the previous tile's last reads of buffer 0)
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (!(MODE & 8) && !(MODE & 128) && kt + 1 < nk) gload(kt + 1);
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        f32x4 a[2], b[2];
        if (MODE & 32) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            a[i] = f32x4{ra[i].x, ra[i].y, ra[i].z, ra[i].w};
            b[i] = f32x4{rw[i].x, rw[i].y, rw[i].z, rw[i].w};
            asm volatile("" : "+v"(a[i]), "+v"(b[i]));
          }
        } else {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(&As[buf][wm * 64 + mi * 32 + frow][kg * 8 + fk]);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(&Ws[buf][wn * 64 + ni * 32 + frow][kg * 8 + fk]);
        }
        if ((MODE & 512) && kg == 1) {
          __builtin_amdgcn_sched_barrier(0);
          if (kt + 1 < nk) lstore(buf ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
      }
      if (MODE & 64) {
        if (kt + 1 < nk) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < 2; ++i)
            asm volatile("" ::"v"(ra[i].x), "v"(ra[i].y), "v"(ra[i].z), "v"(ra[i].w), "v"(rw[i].x), "v"(rw[i].y), "v"(rw[i].z), "v"(rw[i].w));
        }
      } else if (!(MODE & 8) && !(MODE & 512) && kt + 1 < nk) {
        lstore(buf ^ 1);
      }
      if (!(MODE & 16)) __syncthreads();
    }
    const int col_in = lane & 31, row_hi = (lane >> 5) * 4;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + col_in;
        if (n >= e.N) continue;
        const float bv = e.bias ? e.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + row_hi;
          if (m >= e.M) continue;
          float v = acc[mi][ni][r] + bv;
          if (e.relu) v = fmaxf(v, 0.f);
          if (MODE & 4) {
            if (v == 1234.5678f) e.C[(size_t)m * e.ldc + n] = v;
          } else {
            if (e.res) v += e.res[(size_t)m * e.ldres + n];
            e.C[(size_t)m * e.ldc + n] = v;
          }
        }
      }
  }
}

// Distance-2 operand prefetch: the global loads of k-tile kt + 2 are issued at the top of iteration kt into a second
// register set, the set that arrived during iteration kt - 1 goes to LDS right away (its ds_writes overlap the MFMAs),
// and the iteration ends with the barrier alone.  +16 VGPRs: OCC workgroups per CU.
template <int OCC, int PERS>
__global__ __launch_bounds__(256, OCC) void g32_d2(const float* __restrict__ A, const float* __restrict__ W, int lda, int ldw,
                                                   int K, int tiles_n, int n_tiles, Epi e) {
  __shared__ __attribute__((aligned(16))) float As[2][128][LD];
  __shared__ __attribute__((aligned(16))) float Ws[2][128][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 2, lc4 = (tid & 3) * 4;
  const int frow = lane & 31, fk = (lane >> 5) * 4;
  const int nk = K / BK;
  for (int t = blockIdx.x; t < n_tiles; t += PERS ? gridDim.x : n_tiles) {
    const int tile = xcd_remap(t, n_tiles);
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
    f32x4 ra0_0, ra0_1, rw0_0, rw0_1, ra1_0, ra1_1, rw1_0, rw1_1;
    // rows past M / N are clamped: their products are never stored
    const float* ap0 = A + (size_t)min(m0 + lrow, e.M - 1) * lda + lc4;
    const float* ap1 = A + (size_t)min(m0 + lrow + 64, e.M - 1) * lda + lc4;
    const float* wp0 = W + (size_t)min(n0 + lrow, e.N - 1) * ldw + lc4;
    const float* wp1 = W + (size_t)min(n0 + lrow + 64, e.N - 1) * ldw + lc4;
#define G32_GLOAD(kt, RA, RW)                                        \
  do {                                                               \
    RA##_0 = *reinterpret_cast<const f32x4*>(ap0 + (kt) * BK);       \
    RA##_1 = *reinterpret_cast<const f32x4*>(ap1 + (kt) * BK);       \
    RW##_0 = *reinterpret_cast<const f32x4*>(wp0 + (kt) * BK);       \
    RW##_1 = *reinterpret_cast<const f32x4*>(wp1 + (kt) * BK);       \
  } while (0)
#define G32_LSTORE(buf, RA, RW)                                            \
  do {                                                                     \
    *reinterpret_cast<f32x4*>(&As[buf][lrow][lc4]) = RA##_0;               \
    *reinterpret_cast<f32x4*>(&As[buf][lrow + 64][lc4]) = RA##_1;          \
    *reinterpret_cast<f32x4*>(&Ws[buf][lrow][lc4]) = RW##_0;               \
    *reinterpret_cast<f32x4*>(&Ws[buf][lrow + 64][lc4]) = RW##_1;          \
  } while (0)
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    G32_GLOAD(0, ra0, rw0);
    if (nk > 1) G32_GLOAD(1, ra1, rw1);
    if (PERS) __syncthreads();
    G32_LSTORE(0, ra0, rw0);
    __syncthreads();
    auto mfmas = [&](int buf) {
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(&As[buf][wm * 64 + mi * 32 + frow][kg * 8 + fk]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(&Ws[buf][wn * 64 + ni * 32 + frow][kg * 8 + fk]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
      }
    };
    for (int kt = 0; kt < nk; kt += 2) {
      // even iteration: LDS buffer 0 holds tile kt; set 1 holds tile kt + 1 (arrived during the previous iteration)
      if (kt + 1 < nk) G32_LSTORE(1, ra1, rw1);
      if (kt + 2 < nk) G32_GLOAD(kt + 2, ra0, rw0);
      mfmas(0);
      __syncthreads();
      if (kt + 1 < nk) {
        if (kt + 2 < nk) G32_LSTORE(0, ra0, rw0);
        if (kt + 3 < nk) G32_GLOAD(kt + 3, ra1, rw1);
        mfmas(1);
        __syncthreads();
      }
    }
    const int col_in = lane & 31, row_hi = (lane >> 5) * 4;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + col_in;
        if (n >= e.N) continue;
        const float bv = e.bias ? e.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + row_hi;
          if (m >= e.M) continue;
          float v = acc[mi][ni][r] + bv;
          if (e.relu) v = fmaxf(v, 0.f);
          if (e.res) v += e.res[(size_t)m * e.ldres + n];
          e.C[(size_t)m * e.ldc + n] = v;
        }
      }
  }
}

// Operands by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write, no VGPR round trip.  One wave
// instruction fills 1 KiB of LDS linearly; WHICH (row, 16-byte k segment) lands in a slot is the lane's choice of global
// address, so the swizzle is free: slot(r, s) = 4 r + (s ^ ((r >> 1) & 3)) serves a fragment ds_read_b128 in
// two passes without padding (LDS 32 KB per workgroup); the conflict-free XOR ((r >> 2) & 3) measured 0.5 % slower in the
// product (profiles/r03_call29_30_gemm32_swizzle_ab.txt).
template <int PERS>
__global__ __launch_bounds__(256, 4) void g32_dma(const float* __restrict__ A, const float* __restrict__ W, int lda, int ldw,
                                                  int K, int tiles_n, int n_tiles, Epi e) {
  __shared__ __attribute__((aligned(1024))) float L[2][2][128 * BK];  // [buffer][A | W][slots]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, hi = lane >> 5;
  const int nk = K / BK;
  const unsigned lds0 = (unsigned)(size_t)&L[0][0][0];
  // fragment read offsets (bytes inside a [128 x 16] operand image): row base + swizzled segment of k group 0 / 1
  const int xr = (frow >> 1) & 3;
  const unsigned fa0 = (unsigned)((wm * 64 + frow) * 64 + 16 * (hi ^ xr)), fa1 = fa0 ^ 32u;
  const unsigned fb0 = (unsigned)((wn * 64 + frow) * 64 + 16 * (hi ^ xr)), fb1 = fb0 ^ 32u;
  for (int t = blockIdx.x; t < n_tiles; t += PERS ? gridDim.x : n_tiles) {
    const int tile = xcd_remap(t, n_tiles);
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
    // DMA source offsets of this wave's two instructions per operand: slot 64 i + lane -> (row, segment)
    unsigned va[2], vw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = wave + 4 * j, r = 16 * i + (lane >> 2), sg = (lane & 3) ^ ((r >> 1) & 3);
      va[j] = (unsigned)((min(m0 + r, e.M - 1) * lda + sg * 4) * 4);
      vw[j] = (unsigned)((min(n0 + r, e.N - 1) * ldw + sg * 4) * 4);
    }
    auto dma = [&](int kt, int buf) {
      const char* ga = reinterpret_cast<const char*>(A + kt * BK);
      const char* gw = reinterpret_cast<const char*>(W + kt * BK);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned la = lds0 + (unsigned)(buf * 2 * 128 * BK * 4) + (unsigned)((wave + 4 * j) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(va[j]), "s"(ga), "s"(la) : "memory");
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vw[j]), "s"(gw), "s"(la + 128 * BK * 4)
                     : "memory");
      }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto mfmas = [&](int buf) {
      const char* la = reinterpret_cast<const char*>(&L[buf][0][0]);
      const char* lb = reinterpret_cast<const char*>(&L[buf][1][0]);
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(la + (kg ? fa1 : fa0) + mi * 32 * 64);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(lb + (kg ? fb1 : fb0) + ni * 32 * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
      }
    };
    for (int kt = 0; kt < nk; kt += 2) {
      if (kt + 1 < nk) dma(kt + 1, 1);
      mfmas(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < nk) {
        if (kt + 2 < nk) dma(kt + 2, 0);
        mfmas(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    const int col_in = lane & 31, row_hi = (lane >> 5) * 4;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + col_in;
        if (n >= e.N) continue;
        const float bv = e.bias ? e.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + row_hi;
          if (m >= e.M) continue;
          float v = acc[mi][ni][r] + bv;
          if (e.relu) v = fmaxf(v, 0.f);
          if (e.res) v += e.res[(size_t)m * e.ldres + n];
          e.C[(size_t)m * e.ldc + n] = v;
        }
      }
  }
}

// The DMA form with a ring of three operand buffers (48 KB: 3 workgroups per CU): k-tile kt + 2 is requested at the top
// of iteration kt, and the iteration ends by waiting only for k-tile kt + 1 (vmcnt(4): this wave's four newer requests may
// stay in flight).
template <int PERS>
__global__ __launch_bounds__(256, 3) void g32_dma3(const float* __restrict__ A, const float* __restrict__ W, int lda, int ldw,
                                                   int K, int tiles_n, int n_tiles, Epi e) {
  __shared__ __attribute__((aligned(1024))) float L[3][2][128 * BK];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, hi = lane >> 5;
  const int nk = K / BK;
  const unsigned lds0 = (unsigned)(size_t)&L[0][0][0];
  const int xr = (frow >> 1) & 3;
  const unsigned fa0 = (unsigned)((wm * 64 + frow) * 64 + 16 * (hi ^ xr)), fa1 = fa0 ^ 32u;
  const unsigned fb0 = (unsigned)((wn * 64 + frow) * 64 + 16 * (hi ^ xr)), fb1 = fb0 ^ 32u;
  for (int t = blockIdx.x; t < n_tiles; t += PERS ? gridDim.x : n_tiles) {
    const int tile = xcd_remap(t, n_tiles);
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
    unsigned va[2], vw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = wave + 4 * j, r = 16 * i + (lane >> 2), sg = (lane & 3) ^ ((r >> 1) & 3);
      va[j] = (unsigned)((min(m0 + r, e.M - 1) * lda + sg * 4) * 4);
      vw[j] = (unsigned)((min(n0 + r, e.N - 1) * ldw + sg * 4) * 4);
    }
    auto dma = [&](int kt, int buf) {
      const char* ga = reinterpret_cast<const char*>(A + kt * BK);
      const char* gw = reinterpret_cast<const char*>(W + kt * BK);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned la = lds0 + (unsigned)(buf * 2 * 128 * BK * 4) + (unsigned)((wave + 4 * j) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(va[j]), "s"(ga), "s"(la) : "memory");
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vw[j]), "s"(gw), "s"(la + 128 * BK * 4)
                     : "memory");
      }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    dma(0, 0);
    if (nk > 1) dma(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto mfmas = [&](int buf) {
      const char* la = reinterpret_cast<const char*>(&L[buf][0][0]);
      const char* lb = reinterpret_cast<const char*>(&L[buf][1][0]);
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(la + (kg ? fa1 : fa0) + mi * 32 * 64);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(lb + (kg ? fb1 : fb0) + ni * 32 * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
      }
    };
    // iteration kt: request tile kt + 2 into buffer (kt + 2) % 3 (last read in iteration kt - 1: free since its barrier),
    // compute on buffer kt % 3, then wait for tile kt + 1 (all but this wave's 4 newest requests) and meet
    auto step = [&](int kt, int b0, int b2) {
      if (kt + 2 < nk) dma(kt + 2, b2);
      mfmas(b0);
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 3) {
      step(kt, 0, 2);
      if (kt + 1 < nk) step(kt + 1, 1, 0);
      if (kt + 2 < nk) step(kt + 2, 2, 1);
    }
    const int col_in = lane & 31, row_hi = (lane >> 5) * 4;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + col_in;
        if (n >= e.N) continue;
        const float bv = e.bias ? e.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + row_hi;
          if (m >= e.M) continue;
          float v = acc[mi][ni][r] + bv;
          if (e.relu) v = fmaxf(v, 0.f);
          if (e.res) v += e.res[(size_t)m * e.ldres + n];
          e.C[(size_t)m * e.ldc + n] = v;
        }
      }
  }
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t err_ = (x);                                                     \
    if (err_ != hipSuccess) {                                                  \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(err_), __FILE__, __LINE__); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

int main() {
  const int M = 32000;
  struct Shape { const char* name; int N, K, relu, res; } shapes[] = {
      {"qkv     ", 1392, 464, 0, 0}, {"attn_out", 464, 464, 0, 1}, {"ffn1    ", 1856, 464, 1, 0}, {"ffn2    ", 464, 1856, 0, 1}};
  float *A, *W, *C, *R, *bias;
  CK(hipMalloc(&A, (size_t)M * 1856 * 4));
  CK(hipMalloc(&W, (size_t)1856 * 1856 * 4));
  CK(hipMalloc(&C, (size_t)M * 1856 * 4));
  CK(hipMalloc(&R, (size_t)M * 1856 * 4));
  CK(hipMalloc(&bias, 1856 * 4));
  {
    std::vector<float> h((size_t)M * 1856);
    unsigned s = 1u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.1f; }
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(R, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)1856 * 1856 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), 1856 * 4, hipMemcpyHostToDevice));
  }
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, g32<0>, 256, 0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# occupancy reported by the runtime: %d workgroups per CU, %d CUs\n", occ, prop.multiProcessorCount);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int G = 4 * prop.multiProcessorCount;
  for (const Shape& sh : shapes) {
    const int tiles_m = (M + 127) / 128, tiles_n = (sh.N + 127) / 128, n_tiles = tiles_m * tiles_n;
    const double flop = 2.0 * M * sh.N * sh.K;
    const double mfma_us = (double)n_tiles / prop.multiProcessorCount * (sh.K / BK) * 2048.0 / 2400.0;  // at 2.4 GHz
    printf("%s M=%d N=%d K=%d: %d tiles (%.2f rounds of %d), MFMA-bound %.0f us at 2.4 GHz\n", sh.name, M, sh.N, sh.K, n_tiles,
           (double)n_tiles / G, G, mfma_us);
    auto run = [&](const char* label, int mode, int grid, int Kx, int stag) -> int {
      Epi e{bias, sh.res ? R : nullptr, C, M, sh.N, sh.N, sh.N, sh.relu};
      auto launch = [&]() {
        switch (mode) {
          case 0: hipLaunchKernelGGL(g32<0>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 1: hipLaunchKernelGGL(g32<1>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 3: hipLaunchKernelGGL(g32<3>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 4: hipLaunchKernelGGL(g32<4>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 12: hipLaunchKernelGGL(g32<12>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 28: hipLaunchKernelGGL(g32<28>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 60: hipLaunchKernelGGL(g32<60>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 68: hipLaunchKernelGGL(g32<68>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 132: hipLaunchKernelGGL(g32<132>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 260: hipLaunchKernelGGL(g32<260>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 516: hipLaunchKernelGGL(g32<516>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 512: hipLaunchKernelGGL(g32<512>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 513: hipLaunchKernelGGL(g32<513>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
          case 2003: hipLaunchKernelGGL((g32_d2<3, 0>), dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, e); break;
          case 2013: hipLaunchKernelGGL((g32_d2<3, 1>), dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, e); break;
          case 2004: hipLaunchKernelGGL((g32_d2<4, 0>), dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, e); break;
          case 2014: hipLaunchKernelGGL((g32_d2<4, 1>), dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, e); break;
          case 3000: hipLaunchKernelGGL((g32_dma<0>), dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, e); break;
          case 3001: hipLaunchKernelGGL((g32_dma<1>), dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, e); break;
          case 3100: hipLaunchKernelGGL((g32_dma3<0>), dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, e); break;
          case 3101: hipLaunchKernelGGL((g32_dma3<1>), dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, e); break;
          case 20: hipLaunchKernelGGL(g32<20>, dim3(grid), dim3(256), 0, 0, A, W, sh.K, sh.K, Kx, tiles_n, n_tiles, stag, e); break;
        }
      };
      for (int i = 0; i < 15; ++i) launch();
      CK(hipEventRecord(e0, 0));
      const int it = 30;
      for (int i = 0; i < it; ++i) launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / it;
      printf("   %-28s %8.1f us  %6.1f TFLOP/s\n", label, us, flop * ((double)Kx / sh.K) / us * 1e-6);
      return 0;
    };
    const int tile_cycles = (sh.K / BK) * 2048;  // one wave's MFMA time of a tile = a quarter of a tile's wall time at 4 per CU
    if (run("base", 0, n_tiles, sh.K, 0)) return 1;
    if (run("noepi (ablation)", 4, n_tiles, sh.K, 0)) return 1;
    if (run("k1: one k-tile (ablation)", 0, n_tiles, BK, 0)) return 1;
    if (run("noepi, no staging in loop", 12, n_tiles, sh.K, 0)) return 1;
    if (run("noepi, no staging, no barrier", 28, n_tiles, sh.K, 0)) return 1;
    if (run("noepi, MFMA only", 60, n_tiles, sh.K, 0)) return 1;
    if (run("noepi, A rows wrap at 256 (L2-resident)", 260, n_tiles, sh.K, 0)) return 1;
    if (run("noepi, LDS writes mid-iteration", 516, n_tiles, sh.K, 0)) return 1;
    if (run("LDS writes mid-iteration", 512, n_tiles, sh.K, 0)) return 1;
    if (run("pers, LDS writes mid-iteration", 513, G, sh.K, 0)) return 1;
    {  // the DMA form computes the same product (same k order per MFMA group: bit-identical)
      std::vector<float> c0((size_t)M * sh.N), c1((size_t)M * sh.N);
      Epi e{bias, sh.res ? R : nullptr, C, M, sh.N, sh.N, sh.N, sh.relu};
      hipLaunchKernelGGL(g32<0>, dim3(n_tiles), dim3(256), 0, 0, A, W, sh.K, sh.K, sh.K, tiles_n, n_tiles, 0, e);
      CK(hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemset(C, 0, c0.size() * 4));
      hipLaunchKernelGGL((g32_dma<1>), dim3(G), dim3(256), 0, 0, A, W, sh.K, sh.K, sh.K, tiles_n, n_tiles, e);
      CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost));
      size_t bad = 0;
      double mx = 0;
      for (size_t i = 0; i < c0.size(); ++i) {
        const double d = fabs((double)c0[i] - (double)c1[i]);
        if (d > mx) mx = d;
        if (c0[i] != c1[i]) ++bad;
      }
      printf("   check: LDS-DMA (persistent) vs base: %zu of %zu elements differ, max |diff| %.3g\n", bad, c0.size(), mx);
    }
    if (run("LDS-DMA operands", 3000, n_tiles, sh.K, 0)) return 1;
    if (run("LDS-DMA operands, pers", 3001, G, sh.K, 0)) return 1;
    if (run("LDS-DMA operands, pers, 5 per CU", 3001, 5 * prop.multiProcessorCount, sh.K, 0)) return 1;
    if (run("LDS-DMA ring of 3 buffers, 3 per CU", 3100, n_tiles, sh.K, 0)) return 1;
    if (run("LDS-DMA ring of 3, pers, 3 per CU", 3101, 3 * prop.multiProcessorCount, sh.K, 0)) return 1;
    {  // the ring computes the same product
      std::vector<float> c0((size_t)M * sh.N), c1((size_t)M * sh.N);
      Epi e{bias, sh.res ? R : nullptr, C, M, sh.N, sh.N, sh.N, sh.relu};
      hipLaunchKernelGGL(g32<0>, dim3(n_tiles), dim3(256), 0, 0, A, W, sh.K, sh.K, sh.K, tiles_n, n_tiles, 0, e);
      CK(hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemset(C, 0, c0.size() * 4));
      hipLaunchKernelGGL((g32_dma3<1>), dim3(3 * prop.multiProcessorCount), dim3(256), 0, 0, A, W, sh.K, sh.K, sh.K, tiles_n, n_tiles, e);
      CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < c0.size(); ++i) bad += c0[i] != c1[i];
      printf("   check: LDS-DMA ring of 3 (persistent) vs base: %zu of %zu elements differ\n", bad, c0.size());
    }
    if (run("distance-2 prefetch, 3 per CU", 2003, n_tiles, sh.K, 0)) return 1;
    if (run("distance-2 prefetch, 3 per CU, pers", 2013, 3 * prop.multiProcessorCount, sh.K, 0)) return 1;
    if (run("distance-2 prefetch, 4 per CU (spills?)", 2004, n_tiles, sh.K, 0)) return 1;
    if (run("distance-2 prefetch, 4 per CU, pers", 2014, G, sh.K, 0)) return 1;
    if (run("noepi, global loads, no LDS writes", 68, n_tiles, sh.K, 0)) return 1;
    if (run("noepi, LDS writes, no global loads", 132, n_tiles, sh.K, 0)) return 1;
    if (run("pers G=1024", 1, G, sh.K, 0)) return 1;
    (void)tile_cycles;
  }
  return 0;
}
