// MFMA GEMMs for the denoiser's five Linear classes (QKV, attn-out, FFN1, FFN2, vocab head):
//     C[M,N] = epilogue(A[M,K] · W[N,K]^T + bias[N])         (torch.nn.Linear convention, "NT")
// Reference call sites: torch.nn.MultiheadAttention in/out projections
// (trainer/models/transformer_utils.py:140-142,197-204), linear1/linear2 (l.145-147,208-209),
// head Linear (trainer/models/common/nn_lib.py:186-189).
//
// gfx950 design notes
//  * 64-lane wavefronts; a 256-thread workgroup = 4 waves in a 2x2 grid, each wave owns a 64x64
//    sub-tile = 2x2 MFMA 32x32 accumulators (64 accumulator VGPRs).
//  * exact mode: v_mfma_f32_32x32x2_f32 (bit-exact fmaf chain, 157 TF peak).  Lane l supplies
//    A[i=l&31][k=l>>5]; we let each lane read 4 consecutive k (one ds_read_b128) and feed 4 MFMAs,
//    i.e. MFMA j of a group contracts k = {k0+j, k0+4+j} — a permutation of the K order that is
//    applied to A and W alike, so the product is unchanged.
//  * fp16 modes: v_mfma_f32_32x32x16_f16, lane l supplies 8 consecutive k (ds_read_b128).
//  * exact mode operands arrive by LDS-DMA into unpadded, XOR-swizzled images (see gemm_f32_tile); the fp16 kernels of the
//    generic path stage global -> register -> LDS, double-buffered, with rows padded by 8 halfs.
//  * workgroup -> tile mapping is XCD-aware: the 8 XCDs have private L2s and the dispatcher places
//    block b on XCD b%8, so consecutive tile ids (same A row-panel, neighbouring W panels) are
//    remapped onto the same XCD.
#include <cstdlib>

#include "ldm_kernels.h"

namespace ldm {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr float kLoScaleInv = 1.0f / kSplitLoScale;
constexpr float kLoScale = kSplitLoScale;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective for any nwg (cdna guide §5 "XCD swizzle must be bijective")
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// ---------------------------------------------------------------- shared epilogue
struct EpiArgs {
  const float* bias;
  const float* res;
  float* C32;
  __half* C16;
  __half* C16lo;
  int M, N, ldres, ldc32, ldc16, relu;
  float out_scale = 1.0f;  // split mode: undoes the weight tensor's power-of-two pre-scale (applied to the accumulators)
};

template <int MI = 2>
__device__ __forceinline__ void epilogue_store(const EpiArgs& e, const f32x16 (&acc)[MI][2], int m_base, int n_base,
                                               int lane) {
  const int col_in = lane & 31;
  const int row_hi = (lane >> 5) * 4;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n_base + ni * 32 + col_in;
      if (n >= e.N) continue;
      const float bv = e.bias ? e.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_base + mi * 32 + (r & 3) + 8 * (r >> 2) + row_hi;
        if (m >= e.M) continue;
        float v = acc[mi][ni][r] + bv;
        if (e.relu) v = fmaxf(v, 0.f);
        if (e.res) v += e.res[(size_t)m * e.ldres + n];
        if (e.C32) e.C32[(size_t)m * e.ldc32 + n] = v;
        if (e.C16) {
          const __half h = __float2half_rn(v);
          e.C16[(size_t)m * e.ldc16 + n] = h;
          if (e.C16lo) e.C16lo[(size_t)m * e.ldc16 + n] = __float2half_rn((v - __half2float(h)) * kLoScale);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- exact fp32
constexpr int F32_BK = 16;

// Operands by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write, no VGPR round trip.  One wave
// instruction fills 1 KiB of LDS linearly, and WHICH (row, 16-byte k segment) lands in a slot is the lane's choice of
// global address, so the swizzle is free: slot(r, s) = 4 r + (s ^ ((r >> 1) & 3)) serves a fragment
// ds_read_b128 in two passes without padding (32 KB of LDS per workgroup).  The XOR with (r >> 2) & 3 is conflict-free
// under gfx950's 16-lane service groups (tests/test_lds_swizzle.py) and was measured: 0.5 % SLOWER in the timed loop, FFN1 and
// in_proj 2-4 % slower run alone (profiles/r03_call29_30_gemm32_swizzle_ab.txt): the LDS is far from saturated by
// this kernel (12 fragment reads per 32 MFMAs of 64 cycles), so the reads' pace is not what limits it.
// Persistent over tiles: the grid is min(tiles, 4 per CU) workgroups and workgroup w walks tiles w, w + grid, ... (the
// same XCD: grid is a multiple of 8); a workgroup that moves on leaves its 64 epilogue stores to drain from L2 under the
// next main loop and pays the launch / first-load ramp once.  tools/microbench/gemm32.hip on the four shapes of a layer
// (M = 32 000): register-staged, one workgroup per tile 414 / 161 / 507 / 527 us -> 334 / 155 / 446 / 475 us, bit-identical
// (profiles/r03_gemm32_microbench.txt).
// MI = 2: 128 x 128 tile (each wave 64 x 64).  MI = 1: 64 x 128 tile (each wave 32 x 64; 24 KB of LDS) for the GEMMs whose
// 128-row tiles fit the chip in one round: with two half-height tiles per workgroup the stores of the first drain under the
// main loop of the second (opt-in, LDM_GEMM32_BM64=1).
template <int MI>
__global__ __launch_bounds__(256, 4) void gemm_f32_tile(const float* __restrict__ A, const float* __restrict__ W,
                                                        int lda, int ldw, int K, int tiles_n, int n_tiles, EpiArgs e) {
  constexpr int BM = 64 * MI;
  constexpr int kImgA = BM * F32_BK * 4, kImgW = 128 * F32_BK * 4;  // bytes
  __shared__ __attribute__((aligned(1024))) char L[2][kImgA + kImgW];  // [buffer][A image | W image]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, hi = lane >> 5;
  const int nk = K / F32_BK;
  const unsigned lds0 = (unsigned)(size_t)&L[0][0];
  // fragment read offsets (bytes inside a [rows x 16] operand image): row base + swizzled segment of k group 0 / 1
  const int xr = (frow >> 1) & 3;
  const unsigned fa0 = (unsigned)((wm * 32 * MI + frow) * 64 + 16 * (hi ^ xr)), fa1 = fa0 ^ 32u;
  const unsigned fb0 = (unsigned)(kImgA + (wn * 64 + frow) * 64 + 16 * (hi ^ xr)), fb1 = fb0 ^ 32u;
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int tile = xcd_remap(t, n_tiles);
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * 128;
    const float* At = A + (size_t)m0 * lda;
    const float* Wt = W + (size_t)n0 * ldw;
    // DMA source offsets (bytes from the tile's first row) of this wave's instructions: LDS slot 64 i + lane <- (row,
    // segment); A image: instructions w (+ 4 for MI = 2), W image: w and w + 4.  Rows past M / N are clamped to the last
    // one: their products are never stored.
    unsigned va[2], vw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = wave + 4 * j, r = 16 * i + (lane >> 2), sg = (lane & 3) ^ ((r >> 1) & 3);
      va[j] = (unsigned)((min(r, e.M - 1 - m0) * lda + sg * 4) * 4);
      vw[j] = (unsigned)((min(r, e.N - 1 - n0) * ldw + sg * 4) * 4);
    }
    auto dma = [&](int kt, int buf) {
      const char* ga = reinterpret_cast<const char*>(At + kt * F32_BK);
      const char* gw = reinterpret_cast<const char*>(Wt + kt * F32_BK);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned la = lds0 + (unsigned)(buf * (kImgA + kImgW)) + (unsigned)((wave + 4 * j) * 1024);
        if (j < MI)
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(va[j]), "s"(ga), "s"(la) : "memory");
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vw[j]), "s"(gw), "s"(la + kImgA)
                     : "memory");
      }
    };
    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    // (the previous tile's last iteration ended with a barrier: both buffers are free)
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // hipcc does not count the DMA in its vmcnt bookkeeping
    __syncthreads();
    auto mfmas = [&](int buf) {
      const char* lb = &L[buf][0];
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        f32x4 a[MI], b[2];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(lb + (kg ? fa1 : fa0) + mi * 32 * 64);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(lb + (kg ? fb1 : fb0) + ni * 32 * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
      }
    };
    for (int kt = 0; kt < nk; kt += 2) {  // (two iterations per trip: the buffer index is static)
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
    epilogue_store<MI>(e, acc, m0 + wm * 32 * MI, n0 + wn * 64, lane);
  }
}

// N = 464 (attn-out, FFN2: 43 % of the exact-mode step): 128-wide N tiles cover it with 4 tiles = 512 columns (10 % of the
// MFMAs multiply padding).  This variant computes a 128 x 160 tile (3 tiles = 480 columns: 3.4 % padding; 750 workgroups,
// one round): the four waves stack along M (32 rows x 160 columns = 5 accumulator tiles each), one A fragment and five W
// fragments per 4-k group.  Operands by LDS-DMA exactly as above (A image 8 KB, W image 10 KB, two buffers: 36 KB).
constexpr int F32_BN2 = 160;

__global__ __launch_bounds__(256, 4) void gemm_f32_128x160(const float* __restrict__ A, const float* __restrict__ W,
                                                        int lda, int ldw, int K, int tiles_n, int n_tiles, EpiArgs e) {
  constexpr int kImgA = 128 * F32_BK * 4, kImgW = F32_BN2 * F32_BK * 4;  // bytes
  __shared__ __attribute__((aligned(1024))) char L[2][kImgA + kImgW];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nk = K / F32_BK;
  const unsigned lds0 = (unsigned)(size_t)&L[0][0];
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    // everything lane-derived is rebuilt per tile from the hardware lane id (80 accumulators + 24 fragment registers
    // leave no room to carry it through the main loop at 4 workgroups per CU)
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int frow = lane & 31, hi = lane >> 5;
    const int xr = (frow >> 1) & 3;
    const unsigned fb0 = (unsigned)(frow * 64 + 16 * (hi ^ xr));  // (A: + wave * 2048; W: + kImgA + ni * 2048; k group 1: ^ 32)
    const int tile = xcd_remap(t, n_tiles);
    const int m0 = (tile / tiles_n) * 128;
    const int n0 = (tile % tiles_n) * F32_BN2;
    const float* At = A + (size_t)m0 * lda;
    const float* Wt = W + (size_t)n0 * ldw;
    // A image: DMA instructions 0..7 (waves w, w + 4); W image: 0..9 (w, w + 4, and 8 + w for w < 2)
    unsigned va[2], vw[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = wave + 4 * j, r = 16 * i + (lane >> 2), sg = (lane & 3) ^ ((r >> 1) & 3);
      if (j < 2) va[j] = (unsigned)((min(r, e.M - 1 - m0) * lda + sg * 4) * 4);
      vw[j] = (unsigned)((min(r, e.N - 1 - n0) * ldw + sg * 4) * 4);
    }
    auto dma = [&](int kt, int buf) {
      const char* ga = reinterpret_cast<const char*>(At + kt * F32_BK);
      const char* gw = reinterpret_cast<const char*>(Wt + kt * F32_BK);
      const unsigned lb = lds0 + (unsigned)(buf * (kImgA + kImgW));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned la = lb + (unsigned)((wave + 4 * j) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(va[j]), "s"(ga), "s"(la) : "memory");
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vw[j]), "s"(gw), "s"(la + kImgA)
                     : "memory");
      }
      if (wave < 2) {
        const unsigned la = lb + (unsigned)(kImgA + (wave + 8) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vw[2]), "s"(gw), "s"(la) : "memory");
      }
    };
    f32x16 acc[5];
#pragma unroll
    for (int ni = 0; ni < 5; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto mfmas = [&](int buf) {
      const char* lb = &L[buf][0];
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        const unsigned fo = kg ? (fb0 ^ 32u) : fb0;
        const f32x4 a = *reinterpret_cast<const f32x4*>(lb + fo + wave * 32 * 64);
#pragma unroll
        for (int ni = 0; ni < 5; ++ni) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(lb + fo + kImgA + ni * 32 * 64);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[ni], 0, 0, 0);
        }
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
    // epilogue (same element order as epilogue_store: D[i = row][j = column])
    const int col_in = lane & 31;
    const int row_hi = (lane >> 5) * 4;
#pragma unroll
    for (int ni = 0; ni < 5; ++ni) {
      const int n = n0 + ni * 32 + col_in;
      if (n >= e.N) continue;
      const float bv = e.bias ? e.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + row_hi;
        if (m >= e.M) continue;
        float v = acc[ni][r] + bv;
        if (e.relu) v = fmaxf(v, 0.f);
        if (e.res) v += e.res[(size_t)m * e.ldres + n];
        if (e.C32) e.C32[(size_t)m * e.ldc32 + n] = v;
        if (e.C16) {
          const __half h = __float2half_rn(v);
          e.C16[(size_t)m * e.ldc16 + n] = h;
          if (e.C16lo) e.C16lo[(size_t)m * e.ldc16 + n] = __float2half_rn((v - __half2float(h)) * kLoScale);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- fp16 operands, fp32 accumulate
// NPASS = 1: fast (A·W).  NPASS = 3: split (Ahi·Whi + 2^-11·(Ahi·Wlo' + Alo'·Whi)), lo' = lo·2^11.
constexpr int F16_BK = 32;
constexpr int F16_LD = F16_BK + 8;  // 40 halfs = 80 B row stride (same conflict-free pattern)

template <int NPASS>
__global__ __launch_bounds__(256) void gemm_f16_128x128(const __half* __restrict__ A, const __half* __restrict__ Alo,
                                                        const __half* __restrict__ W, const __half* __restrict__ Wlo,
                                                        int lda, int ldw, int K, int tiles_n, EpiArgs e) {
  constexpr int NOP = (NPASS == 3) ? 2 : 1;  // hi (+ lo) images per operand
  __shared__ __attribute__((aligned(16))) __half As[2][NOP][128][F16_LD];
  __shared__ __attribute__((aligned(16))) __half Ws[2][NOP][128][F16_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * 128;
  const int n0 = (tile % tiles_n) * 128;

  // staging: tile = 128 rows x 32 halfs = 4 x 16 B per row -> 512 uint4 -> 2 per thread per image
  const int lrow = tid >> 2;
  const int lc8 = (tid & 3) * 8;
  uint4 ra[NOP][2], rw[NOP][2];
  const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + lrow + i * 64;
      const int n = n0 + lrow + i * 64;
      const size_t ao = (size_t)m * lda + kt * F16_BK + lc8;
      const size_t wo = (size_t)n * ldw + kt * F16_BK + lc8;
      ra[0][i] = (m < e.M) ? *reinterpret_cast<const uint4*>(A + ao) : z4;
      rw[0][i] = (n < e.N) ? *reinterpret_cast<const uint4*>(W + wo) : z4;
      if (NOP == 2) {
        ra[NOP - 1][i] = (m < e.M) ? *reinterpret_cast<const uint4*>(Alo + ao) : z4;
        rw[NOP - 1][i] = (n < e.N) ? *reinterpret_cast<const uint4*>(Wlo + wo) : z4;
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int o = 0; o < NOP; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<uint4*>(&As[buf][o][lrow + i * 64][lc8]) = ra[o][i];
        *reinterpret_cast<uint4*>(&Ws[buf][o][lrow + i * 64][lc8]) = rw[o][i];
      }
  };

  f32x16 acc[2][2], acl[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[mi][ni][r] = 0.f;
        acl[mi][ni][r] = 0.f;
      }

  const int nk = K / F16_BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int frow = lane & 31;
  const int fk = (lane >> 5) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < F16_BK / 16; ++ks) {
      f16x8 a[NOP][2], b[NOP][2];
#pragma unroll
      for (int o = 0; o < NOP; ++o) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          a[o][mi] = *reinterpret_cast<const f16x8*>(&As[buf][o][wm * 64 + mi * 32 + frow][ks * 16 + fk]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          b[o][ni] = *reinterpret_cast<const f16x8*>(&Ws[buf][o][wn * 64 + ni * 32 + frow][ks * 16 + fk]);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
          if (NPASS == 3) {
            acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mi], b[NOP - 1][ni], acl[mi][ni], 0, 0, 0);
            acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[NOP - 1][mi], b[0][ni], acl[mi][ni], 0, 0, 0);
          }
        }
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }
  if (NPASS == 3) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = (acc[mi][ni][r] + acl[mi][ni][r] * kLoScaleInv) * e.out_scale;
  }
  epilogue_store(e, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

void launch_gemm(const GemmArgs& g, hipStream_t st) {
  const int tiles_m = (g.M + 127) / 128;
  const int tiles_n = (g.N + 127) / 128;
  EpiArgs e{g.bias, g.res, g.C32, g.C16, g.C16lo, g.M, g.N, g.ldres, g.ldc32, g.ldc16, g.relu};
  e.out_scale = g.out_scale > 0.f ? g.out_scale : 1.0f;
  dim3 grid(tiles_m * tiles_n), block(256);
  if (g.precision == 0) {
    // opt-in (LDM_GEMM32_WIDE=1): run alone, the 160-wide tiles cut FFN2 by 4 % (366 vs 381 ms per 100 steps), but the
    // timed exact-mode loop runs two chunk pipelines concurrently, where the other lane already fills the idle slots, and
    // there they are 0.5 % slower (profiles/r03_call26_gemm32_wide_ab.txt; the same verdict as r02_call35_37_*)
    static const bool wide = knob_int("LDM_GEMM32_WIDE", 0) != 0;
    const int t160 = (g.N + F32_BN2 - 1) / F32_BN2;
    if (wide && t160 * F32_BN2 < tiles_n * 128) {  // fewer padded columns (N = 464: 480 vs 512)
      static const int resident160 = [] {
        int dev = 0, cus = 256, per_cu = 4;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_f32_128x160, 256, 0) != hipSuccess || per_cu < 1) per_cu = 3;
        return (per_cu * cus) & ~7;
      }();
      const int n_tiles = tiles_m * t160;
      hipLaunchKernelGGL(gemm_f32_128x160, dim3(n_tiles < resident160 ? n_tiles : resident160), block, 0, st, (const float*)g.A,
                         (const float*)g.W, g.lda, g.ldw, g.K, t160, n_tiles, e);
      return;
    }
    static const auto resident_of = [](const void* kern) {  // as many workgroups as the chip holds at once
      int dev = 0, cus = 256, per_cu = 4;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
      if (const char* v = knob_env("LDM_GEMM32_SLOTS")) per_cu = atoi(v) > 0 ? atoi(v) : per_cu;  // (A/B timing)
      return (per_cu * cus) & ~7;  // (a multiple of 8: tile t stays on XCD t % 8)
    };
    static const int resident = resident_of((const void*)gemm_f32_tile<2>);  // 32 KB of LDS, < 100 VGPRs: 5 per CU
    const int n_tiles = tiles_m * tiles_n;
    static const bool bm64 = knob_int("LDM_GEMM32_BM64", 0) != 0;
    if (bm64 && n_tiles <= resident) {  // one round of 128-row tiles: two half-height tiles per workgroup instead
      static const int resident64 = resident_of((const void*)gemm_f32_tile<1>);
      const int n64 = ((g.M + 63) / 64) * tiles_n;
      hipLaunchKernelGGL(gemm_f32_tile<1>, dim3(n64 < resident64 ? n64 : resident64), block, 0, st, (const float*)g.A,
                         (const float*)g.W, g.lda, g.ldw, g.K, tiles_n, n64, e);
      return;
    }
    hipLaunchKernelGGL(gemm_f32_tile<2>, dim3(n_tiles < resident ? n_tiles : resident), block, 0, st, (const float*)g.A,
                       (const float*)g.W, g.lda, g.ldw, g.K, tiles_n, n_tiles, e);
  } else if (g.precision == 1) {
    hipLaunchKernelGGL(gemm_f16_128x128<1>, grid, block, 0, st, (const __half*)g.A, (const __half*)nullptr,
                       (const __half*)g.W, (const __half*)nullptr, g.lda, g.ldw, g.K, tiles_n, e);
  } else {
    hipLaunchKernelGGL(gemm_f16_128x128<3>, grid, block, 0, st, (const __half*)g.A, (const __half*)g.Alo,
                       (const __half*)g.W, (const __half*)g.Wlo, g.lda, g.ldw, g.K, tiles_n, e);
  }
}

}  // namespace ldm
