// Device-side building blocks shared by the row-stationary kernels (kernels_rowgemm.hip, kernels_fusedattn.hip,
// kernels_layer.hip): hand-issued LDS reads with counted waits, the weight-tile pipeline of the attention block and
// the continuous chunk pipeline of the fused FFN.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "ldm_dma.h"
#include "ldm_kernels.h"

namespace ldm {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int RK = 512;              // padded K of the row-stationary operand (halfs)
constexpr int RKB = RK * 2;          // bytes per weight row = one 1-KiB DMA instruction
constexpr int W1_STAGE = 32 * RKB;   // 32 weight rows of 1 KiB
constexpr int TILE_STAGE = 32 * RKB; // one 32-row weight tile of the attention block (32 KiB)
constexpr int FFN_STAGE = 65536;     // one fused-FFN chunk: W1 tile | W2 slab
constexpr int LN_DP = 512;           // LDS parameter image: multiplier at [0, LN_DP), shift at [LN_DP, 2*LN_DP)
constexpr int KV_BYTES = 128 * 128;  // Ks: 128 keys x 64 halfs ; Vs: 64 d x 128 key-slots (both 16 KiB)

typedef __attribute__((address_space(3))) char* lds_char_ptr;

template <int OFF>
__device__ __forceinline__ void dsr128(f16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// a 4-dword fragment parked in the AGPR half of the register file (written once, read much later as an MFMA operand)
__device__ __forceinline__ f16x8 to_agpr4(const f16x8& v) {
  typedef __attribute__((ext_vector_type(4))) float f32x4v;
  const f32x4v in = __builtin_bit_cast(f32x4v, v);
  f32x4v out;
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(out[0]) : "v"(in[0]));
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(out[1]) : "v"(in[1]));
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(out[2]) : "v"(in[2]));
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(out[3]) : "v"(in[3]));
  return __builtin_bit_cast(f16x8, out);
}
__device__ __forceinline__ float to_agpr(float v) {
  float r;
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(v));
  return r;
}
template <int OFF>
__device__ __forceinline__ void dsr128f(float4& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// 29 (KS) dependent-free reads/MFMAs of one weight tile; SWAP = false: acc = W·X^T, true: acc = X·W^T.
// (r02 negative result: splitting the k-steps over TWO accumulator chains — 64 cycles between dependent MFMAs instead
//  of 32 — did not speed the run up (FFN GEMM1: 1350 -> 1287 cycles per 29 MFMAs; here the compiler-managed second
//  accumulator made it slower): the ~45 cycles per MFMA of these runs are not an accumulator-dependency stall.
//  profiles/r02_call6_chains_ab.txt)
template <int KS, int PF>
struct TilePipe {
  f16x8 q[PF];
  unsigned aW[8];
  const f16x8* xf;
  f32x16 acc;
  const char* gnext;  // image of the next tile + wave*8 KiB (uniform)
  unsigned mnext;     // LDS byte address of the next stage + wave*8 KiB (uniform)
  unsigned voff;      // lane*16

  template <int IT>
  __device__ __forceinline__ void read_item() {
    dsr128<256 * (IT >> 3)>(q[IT % PF], aW[IT & 7]);
  }
  // next tile's DMA (linear 32-KiB image per tile, 8 KiB per wave): 1 instruction per slot, M0 rewritten one
  // step ahead of every 4th
  template <int J>
  __device__ __forceinline__ void dma_m0() {
    if constexpr (J < 8 && (J & 3) == 0) dma_set_m0(mnext + (J >> 2) * 4096);
  }
  template <int J>
  __device__ __forceinline__ void dma_slot() {
    if constexpr (J < 8) dma_lin<(J & 3) * 1024>(voff, gnext + (J >> 2) * 4096);
  }
  template <int IT, bool SWAP>
  __device__ __forceinline__ void step() {
    if constexpr (IT < KS) {
      constexpr int after = (KS - 1 - IT) < (PF - 1) ? (KS - 1 - IT) : (PF - 1);
      wait_lgkm<after>();
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 cur = q[IT % PF];
      if constexpr (IT == 0) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[0], cur, zero, 0, 0, 0)
                   : __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, xf[0], zero, 0, 0, 0);
      } else {
        acc = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[IT], cur, acc, 0, 0, 0)
                   : __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, xf[IT], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (IT + PF < KS) read_item<IT + PF>();
      if constexpr (IT % 2 == 1) dma_slot<IT / 2>();
      else dma_m0<IT / 2>();
      step<IT + 1, SWAP>();
    }
  }
  template <int IT>
  __device__ __forceinline__ void prologue() {
    if constexpr (IT < PF) {
      read_item<IT>();
      prologue<IT + 1>();
    }
  }
  template <bool SWAP>
  __device__ __forceinline__ void run() {
    prologue<0>();
    step<0, SWAP>();
  }
};

template <int S8>
__device__ __forceinline__ f16x8 cvt8(const f32x16& a, const float4& b0, const float4& b1) {
  f16x8 o;
  o[0] = (_Float16)(a[S8 + 0] + b0.x); o[1] = (_Float16)(a[S8 + 1] + b0.y);
  o[2] = (_Float16)(a[S8 + 2] + b0.z); o[3] = (_Float16)(a[S8 + 3] + b0.w);
  o[4] = (_Float16)(a[S8 + 4] + b1.x); o[5] = (_Float16)(a[S8 + 5] + b1.y);
  o[6] = (_Float16)(a[S8 + 6] + b1.z); o[7] = (_Float16)(a[S8 + 7] + b1.w);
  return o;
}

// ------------------------------------------------------------------------------------------------
// FfnStream: the chunk loop of the fused FFN as ONE continuous LDS-read / MFMA pipeline.
//
// FfnPipe restarts its read queue at every chunk: barrier, then PF fragment reads whose latency nothing hides, then
// the first MFMA (profiles/r02_call8_ffn_ablation.txt: ~160 cycles at the barrier + ~200 of queue fill per chunk,
// with or without the weight DMA, with or without a rotating B operand).  Here a chunk is NIT = 60 queue items —
// 29 W1 fragments, ONE pseudo item (the bias/ReLU/cast step; it owns a queue slot so that 60 % PF == 0 and item i of
// every chunk lives in slot i % PF), 30 W2 fragments — and the queue never drains: step IT issues the read of item
// IT + PF, which for IT >= NIT - PF is item IT + PF - NIT of the NEXT chunk, in the other LDS stage.
//   * stage (c+1) is complete before those reads: every wave executes s_waitcnt vmcnt(0) (its own DMA pieces, issued
//     at steps 1..31) + s_barrier at step NIT - PF;
//   * stage (c) is not overwritten while it is still read: its last read is ISSUED at step NIT - PF - 1, before that
//     barrier, and the first DMA piece into it is issued at step 1 of chunk c+1 (> 200 cycles later);
//   * the per-lane LDS addresses are toggled between the stages in place (address ^ 0x10000, stage size 64 KiB):
//     aW1 after its last use (step 22), aW2 after its last use (step 53);
//   * the next chunk's bias is read right after the barrier, i.e. older than the next chunk's item 0, so it has
//     landed when chain A's first MFMA takes it as its C operand (counted lgkmcnt waits stay exact: +4 younger
//     operations while waiting for items 55..59).
template <int KS, int NT2, int DE, bool TM>
struct FfnStream {
  static constexpr int PF = 6;
  static constexpr int NIT = KS + 1 + 2 * NT2;  // 60
  static constexpr int SYNC = NIT - PF;         // step whose read is the first of the next chunk
  static_assert(NIT % PF == 0, "queue slots must line up across chunks");
  unsigned long long tA, tB, tC, tD;
  f16x8 q[PF];
  unsigned aW1[8], aW2[2];
  const f16x8* xf;
  f32x16 ha, hb;
  f32x16* acc;
  f16x8 pf[2];
  float4 bb[4];
  static constexpr int IPW = 16;
  const char* gnext;
  unsigned mnext, voff, ab_next;

  template <int J>
  __device__ __forceinline__ void dma_m0() {
    if constexpr (J < IPW && (J & 3) == 0) dma_set_m0(mnext + (J >> 2) * 4096);
  }
  template <int J>
  __device__ __forceinline__ void dma_slot() {
    if constexpr (J < IPW) dma_lin<(J & 3) * 1024>(voff, gnext + (J >> 2) * 4096);
  }
  template <int IT>  // IT in [0, NIT + PF): items >= NIT belong to the next chunk
  __device__ __forceinline__ void read_item() {
    constexpr int I = IT % NIT;
    if constexpr (I < KS) {
      dsr128<256 * (I >> 3)>(q[IT % PF], aW1[I & 7]);
    } else if constexpr (I == KS) {
      dsr128<0>(q[IT % PF], aW1[0]);  // pseudo item (ReLU step): keeps the slot / count bookkeeping uniform
    } else {
      constexpr int sx = (I - KS - 1) / NT2, t = (I - KS - 1) % NT2;
      dsr128<W1_STAGE + t * 2048>(q[IT % PF], aW2[sx]);
    }
  }
  __device__ __forceinline__ void read_bias() {
    dsr128f<0>(bb[0], ab_next);
    dsr128f<32>(bb[1], ab_next);
    dsr128f<64>(bb[2], ab_next);
    dsr128f<96>(bb[3], ab_next);
  }
  template <int IT, bool DMA>
  __device__ __forceinline__ void step() {
    if constexpr (IT < NIT) {
      // LDS operations younger than item IT when it is waited for: PF - 1 items, + the 4 bias reads issued at SYNC
      constexpr int after = (IT > SYNC) ? PF - 1 + 4 : PF - 1;
      wait_lgkm<after>();
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 cur = q[IT % PF];
      if constexpr (IT == 0) {
        f32x16 bv;  // bias in accumulator layout = the C operand of chain A's first MFMA
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          bv[rq * 4 + 0] = bb[rq].x; bv[rq * 4 + 1] = bb[rq].y; bv[rq * 4 + 2] = bb[rq].z; bv[rq * 4 + 3] = bb[rq].w;
        }
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(ha) : "v"(cur), "v"(xf[0]), "v"(bv));
      } else if constexpr (IT == 1) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(hb) : "v"(cur), "v"(xf[1]));
      } else if constexpr (IT < KS) {
        if constexpr (IT % 2 == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ha) : "v"(cur), "v"(xf[IT]));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(hb) : "v"(cur), "v"(xf[IT]));
        if constexpr (IT == KS - 1) asm volatile("s_nop 15" ::: "memory");  // MFMA result -> VALU read (next step)
      } else if constexpr (IT == KS) {
        // bias (already inside chain A) + ReLU + cast: accumulator reg <-> hidden f = (q&3) + 8*(q>>2) + 4*hi
        (void)cur;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          pf[rq >> 1][(rq & 1) * 4 + 0] = (_Float16)fmaxf(ha[rq * 4 + 0] + hb[rq * 4 + 0], 0.f);
          pf[rq >> 1][(rq & 1) * 4 + 1] = (_Float16)fmaxf(ha[rq * 4 + 1] + hb[rq * 4 + 1], 0.f);
          pf[rq >> 1][(rq & 1) * 4 + 2] = (_Float16)fmaxf(ha[rq * 4 + 2] + hb[rq * 4 + 2], 0.f);
          pf[rq >> 1][(rq & 1) * 4 + 3] = (_Float16)fmaxf(ha[rq * 4 + 3] + hb[rq * 4 + 3], 0.f);
        }
      } else {
        constexpr int sx = (IT - KS - 1) / NT2, t = (IT - KS - 1) % NT2;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, pf[sx], acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (TM && IT == KS - 1) tC = __builtin_amdgcn_s_memtime();
      if constexpr (TM && IT == KS + 1) tD = __builtin_amdgcn_s_memtime();
      if constexpr (IT == SYNC) {
        // the next stage is complete (own DMA pieces landed, then everybody's), this stage's reads are all issued
        if constexpr (TM) tA = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (TM) tB = __builtin_amdgcn_s_memtime();
        read_bias();
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) aW2[sx] ^= 0x10000u;  // (last W2 read of this chunk was issued at step SYNC - 1)
      }
      read_item<IT + PF>();
      if constexpr (IT == KS - 1 - PF) {
        // item KS - 1 (the last W1 fragment of this chunk) has just been issued: aW1 now points into the next stage
#pragma unroll
        for (int k = 0; k < 8; ++k) aW1[k] ^= 0x10000u;
      }
      if constexpr (DMA && IT % DE == DE - 1) dma_slot<IT / DE>();
      else if constexpr (DMA && IT % DE == 0) dma_m0<IT / DE>();
      step<IT + 1, DMA>();
    }
  }
  template <int IT>
  __device__ __forceinline__ void prologue() {
    if constexpr (IT < PF) {
      read_item<IT>();
      prologue<IT + 1>();
    }
  }
};

// ------------------------------------------------------------------------------------------------
// SlabPipe: one 32-wide k chunk of the out-projection in K-SLAB form (fused layer kernel): 2 x NT2 MFMAs on NT2
// independent accumulator tiles — A = the slab's rows of output tile t (LDS, W2-slab format of ldm_pack.h),
// B = the chunk's two register-resident attention-output fragments — with the next slab's 32-KiB DMA interleaved.
template <int NT2, int PF>
struct SlabPipe {
  static constexpr int NIT = 2 * NT2;
  f16x8 q[PF];
  unsigned aS[2];
  f32x16* acc;
  const char* gnext;  // image of the next slab + wave*8 KiB (uniform)
  unsigned mnext;     // LDS byte address of the next stage + wave*8 KiB (uniform)
  unsigned voff;      // lane*16

  template <int J>
  __device__ __forceinline__ void dma_m0() {
    if constexpr (J < 8 && (J & 3) == 0) dma_set_m0(mnext + (J >> 2) * 4096);
  }
  template <int J>
  __device__ __forceinline__ void dma_slot() {
    if constexpr (J < 8) dma_lin<(J & 3) * 1024>(voff, gnext + (J >> 2) * 4096);
  }
  template <int IT>
  __device__ __forceinline__ void read_item() {
    constexpr int sx = IT / NT2, t = IT % NT2;
    dsr128<t * 2048>(q[IT % PF], aS[sx]);
  }
  template <int IT>
  __device__ __forceinline__ void step(const f16x8& b0, const f16x8& b1) {
    if constexpr (IT < NIT) {
      constexpr int after = (NIT - 1 - IT) < (PF - 1) ? (NIT - 1 - IT) : (PF - 1);
      wait_lgkm<after>();
      __builtin_amdgcn_sched_barrier(0);
      constexpr int sx = IT / NT2, t = IT % NT2;
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[IT % PF], sx ? b1 : b0, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (IT + PF < NIT) read_item<IT + PF>();
      if constexpr (IT % 2 == 1) dma_slot<IT / 2>();
      else dma_m0<IT / 2>();
      step<IT + 1>(b0, b1);
    }
  }
  template <int IT>
  __device__ __forceinline__ void prologue() {
    if constexpr (IT < PF) {
      read_item<IT>();
      prologue<IT + 1>();
    }
  }
  __device__ __forceinline__ void run(const f16x8& b0, const f16x8& b1) {
    prologue<0>();
    step<0>(b0, b1);
  }
};

}  // namespace ldm
