// Device-side building blocks of the stack kernel (kernels_stack.hip): hand-issued LDS reads with counted waits, the weight-tile pipeline of the attention block and
// the continuous chunk pipeline of the fused FFN.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "ldm_dma.h"
#include "ldm_kernels.h"
#include "ldm_stream_sched.h"

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
// the same, IN PLACE: the new value is written into the AGPRs that hold the old one (read-write operand).  For an array
// element updated under a run-time index (switch over the head): the updated element stays in its registers on every
// path, so the merge after the switch needs no copies (with fresh "=a" outputs hipcc shuffled the whole 128-register
// array through v_accvgpr_mov on every iteration).
__device__ __forceinline__ void park_agpr4(f16x8& dst, const f16x8& v) {
  typedef __attribute__((ext_vector_type(4))) float f32x4v;
  const f32x4v in = __builtin_bit_cast(f32x4v, v);
  f32x4v out = __builtin_bit_cast(f32x4v, dst);
  asm volatile("v_accvgpr_write_b32 %0, %1" : "+a"(out[0]) : "v"(in[0]));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "+a"(out[1]) : "v"(in[1]));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "+a"(out[2]) : "v"(in[2]));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "+a"(out[3]) : "v"(in[3]));
  dst = __builtin_bit_cast(f16x8, out);
}
// scalar form of park_agpr4: overwrite an accumulator element in the AGPR that holds it (a macro: a vector element
// cannot bind to a reference)
#define LDM_SET_AGPR(dst, v) asm volatile("v_accvgpr_write_b32 %0, %1" : "+a"(dst) : "v"(v))
// read an accumulator element through a volatile asm: hipcc otherwise merges the reads of a two-pass loop over the
// accumulators (statistics, then transform) and keeps all 232 values of the first pass alive in VGPRs
__device__ __forceinline__ float get_agpr(float a) {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
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
  template <int IT, bool SWAP, bool DMA = true>
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
      if constexpr (DMA && IT % 2 == 1) dma_slot<IT / 2>();
      else if constexpr (DMA) dma_m0<IT / 2>();
      step<IT + 1, SWAP, DMA>();
    }
  }
  template <int IT>
  __device__ __forceinline__ void prologue() {
    if constexpr (IT < PF) {
      read_item<IT>();
      prologue<IT + 1>();
    }
  }
  template <bool SWAP, bool DMA = true>
  __device__ __forceinline__ void run() {
    prologue<0>();
    step<0, SWAP, DMA>();
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
// (r02 negative result: GEMM1 as ONE accumulator chain in AGPRs — C = 0, bias added in the ReLU step — instead of the
//  two chains in arch VGPRs is 2 % slower, profiles/r02_call28_*: the 44-vs-32 cycles per MFMA of GEMM1 vs GEMM2 are
//  not a VGPR-port effect of VGPR-resident accumulators.)
// PIPE (the stack kernel): SOFTWARE-PIPELINED chunks.  Between the two GEMMs of a chunk the matrix pipe idles: GEMM1's
// result has to drain (MFMA -> VALU hazard), then 24 VALU issues of bias / ReLU / cast, then GEMM2 can start — ~230 of a
// chunk's 2 390 cycles (tools/microbench/dma_feed.hip, profiles/r03_call13_*: 2 387 -> 2 160 cycles per chunk).  With
// PIPE an iteration runs GEMM1 of chunk i and GEMM2 of chunk i - 1 (weights re-timed on the host:
// ldm_pack::pack_ffn_image_pipelined, stage i = W1 tile i | W2 slab i - 1; nc + 1 iterations), and the ReLU / cast of
// chunk i is issued in the shadows of that GEMM2's MFMAs 2..9: pf = fragments of chunk i - 1 (in use), pfn = fragments
// of chunk i (being built), copied at the top of the next iteration.  The LDS traffic, its order and every counted wait
// are unchanged (ldm_stream_sched.h).
template <int KS, int NT2, int DE, bool TM, int PFQ = 6, bool PIPE = false>
struct FfnStream {
  using SCH = ldm_sched::FfnSched<KS, NT2, PFQ>;  // (replayed on the CPU: tests/cpu_sched_check.cpp)
  static constexpr int PF = SCH::PF;   // queue depth.  r02: depth 10 instead of 6 changes nothing (profiles/r02_call21_*)
  static constexpr int NIT = SCH::NIT;   // 60
  static constexpr int SYNC = SCH::SYNC; // step whose read is the first of the next chunk
  unsigned long long tA, tB, tC, tD, t_sync = 0;
  f16x8 q[PF];
  unsigned aW1[8], aW2[2];
  const f16x8* xf;
  f32x16 ha, hb;
  f32x16* acc;
  f16x8 pf[2], pfn[2];
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
      wait_lgkm<SCH::after(IT)>();
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
        if constexpr (IT == KS - 1 && !PIPE) asm volatile("s_nop 15" ::: "memory");  // MFMA result -> VALU read (next step)
      } else if constexpr (IT == KS && PIPE) {
        (void)cur;  // the slot keeps the queue bookkeeping uniform; chunk i's ReLU runs inside GEMM2 below
      } else if constexpr (IT == KS) {
        // bias (already inside chain A) + ReLU + cast: accumulator reg <-> hidden f = (q&3) + 8*(q>>2) + 4*hi
        // Packed: 8 v_pk_add_f32 + 8 v_cvt_pk_f16_f32 + 4 v_pk_max_f16 instead of 16 add + 16 max + 8 cvt — the matrix
        // pipe idles through this step (it consumes GEMM1's result and produces GEMM2's operand), so every VALU issue
        // here is exposed (tools/microbench/dma_feed.hip: 2 270 cycles per chunk with no DMA at all against 59 x 32 = 1 888).
        // ReLU after the rounding: rounding is monotonic and keeps the sign, max(-0, 0) feeds a zero either way.
        (void)cur;
        typedef __attribute__((ext_vector_type(2))) float f32x2v;
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x2v sa = {ha[2 * i], ha[2 * i + 1]}, sb = {hb[2 * i], hb[2 * i + 1]};
          const f16x2v hv = __builtin_convertvector(sa + sb, f16x2v);
          const f16x2v zero = {(_Float16)0.f, (_Float16)0.f};
          const f16x2v r = __builtin_elementwise_max(hv, zero);
          pf[i >> 2][(i & 3) * 2 + 0] = r[0];
          pf[i >> 2][(i & 3) * 2 + 1] = r[1];
        }
      } else {
        constexpr int sx = (IT - KS - 1) / NT2, t = (IT - KS - 1) % NT2;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, pf[sx], acc[t], 0, 0, 0);
        if constexpr (PIPE) {
          // bias (inside chain A) + ReLU + cast of the GEMM1 that finished >= 3 MFMAs (96 cycles) ago, one accumulator
          // pair per MFMA shadow: add, add, cvt_pk, pk_max — scalar adds, packed f32 VALU is slow beside MFMAs
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
      if constexpr (PIPE && IT == 0) {  // the fragments finished during the previous GEMM2 (which no longer reads pf)
        pf[0] = pfn[0];
        pf[1] = pfn[1];
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
        if constexpr (TM) {
          tB = __builtin_amdgcn_s_memtime();
          t_sync += tB - tA;
        }
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
// HeadStream: the six in_proj tiles of ONE head (k0 k1 v0 v1 q0 q1, 29 k-steps each) as ONE
// continuous LDS-read / MFMA pipeline of 174 items.
//
// TilePipe restarts its read queue at every tile (s_barrier, PF exposed fragment reads, 29 MFMAs, then an epilogue
// during which the matrix pipe idles): 175 + 1306 + 439 cycles per tile for 928 cycles of MFMA
// (profiles/r02_call13_*).  Here
//   * the queue never drains inside a head: step G issues the read of item G + PF, which for the last PF steps of a
//     tile is a fragment of the NEXT tile in the next stage of a THREE-stage ring (3 x 32 KiB; stage = tile % 3, a
//     compile-time constant: stages 0 / 1 through the ds_read offset field, stage 2 through a second address set);
//   * the weight DMA runs TWO tiles ahead: tile j + 2 is issued in steps 1..9 of tile j, so a piece has a whole tile
//     (~1 200 cycles) to land instead of ~14 steps (with a two-stage ring the barrier step waited 262 cycles per tile
//     for late pieces: profiles/r02_call16_*);
//   * one s_waitcnt vmcnt(8) + s_barrier per tile at local step KS - PF: every wave has then ISSUED all its reads of
//     this tile (so its stage may be overwritten by tile j + 3, first issued > PF steps later) and its own pieces of
//     tile j + 1 have landed — LDS-DMA pieces retire in order, the 8 youngest are tile j + 2's — then everybody's;
//   * tiles alternate between two accumulators; the epilogue of tile j (fp16 cast + K / V^T ds_write, or the cast
//     into the Q fragments) is issued inside steps EPI0.. of tile j + 1, in the MFMA shadow;
//   * the K / Q bias enters through the C operand of the tile's first MFMA (read at the previous tile's barrier step:
//     older in the LDS queue than the tile's item 0); the V bias is not applied here at all: softmax rows sum to 1,
//     so P·(V + 1 b^T) = P·V + b^T, a constant the host folds into the out-projection bias (b_out + W_out b_v).
// lgkmcnt bookkeeping: every wait is counted exactly from a constexpr replay of the issue order (younger()).
// LEAN (stack kernel, where no AGPR is free to absorb a register peak): the K / Q bias is added in the tile's epilogue
// (read PF + 1 steps ahead of it: the counted wait of the step before the epilogue then covers it) instead of entering through the first MFMA's C operand — no 16-register bias tuple
// beside the two accumulators — and the third ring stage is addressed by a v_add in front of the read instead of a
// second address set.
template <bool TM = false, bool LEAN = false>
struct HeadStream {
  using SCH = ldm_sched::HeadSched<LEAN>;  // the issue schedule (pure constexpr: replayed on the CPU by tests/cpu_sched_check.cpp)
  static constexpr int KS = SCH::KS, NT = SCH::NT, NIT = SCH::NIT, PF = SCH::PF, SYNC = SCH::SYNC, EPI0 = SCH::EPI0;
  static_assert(NT % 3 == 0, "ring stage of a tile is a compile-time constant");
  f16x8 q[PF];
  unsigned aW[8], aW2[8];  // LDS byte addresses of the fragment columns in stage 0 (stage 1: + offset) / stage 2
  const f16x8* xf;      // the wave's 29 activation fragments
  f32x16 accA, accB;    // even / odd tiles
  float4 bb[4];         // bias of the next K / Q tile in accumulator layout (C operand of its first MFMA)
  f16x8* qf;            // [4] Q fragments of this head (B operand of S^T)
  unsigned aK[2], aV[2];  // ds_write addresses of this lane's K row / V^T row pieces (chunk, chunk ^ 2)
  unsigned a_bias;      // LDS byte address of sbias + hi*16
  const char* gimg;     // image of this head's tile 0 + wave * 8 KiB (uniform)
  unsigned lds_w;       // lds0 + wave * 8 KiB (uniform)
  unsigned voff;        // lane * 16
  int h, H;
  unsigned long long t_sync = 0;  // (TM) cycles spent in the per-tile vmcnt + s_barrier

  static constexpr bool tile_has_bias(int j) { return SCH::tile_has_bias(j); }
  static constexpr int younger(int G) { return SCH::younger(G); }

  template <int G>
  __device__ __forceinline__ void read_item() {
    constexpr int IT = G % KS, ST = (G / KS) % 3;
    if constexpr (ST == 2 && LEAN) dsr128<256 * (IT >> 3)>(q[G % PF], aW[IT & 7] + 2u * TILE_STAGE);
    else if constexpr (ST == 2) dsr128<256 * (IT >> 3)>(q[G % PF], aW2[IT & 7]);
    else dsr128<256 * (IT >> 3) + ST * TILE_STAGE>(q[G % PF], aW[IT & 7]);
  }
  // bias of tile j (K: rows H*64.., Q: rows 0..) for head h, d-half t = j & 1 -> bb (assembled into the C operand
  // at the use point, BEHIND the counted wait: a register copy placed next to the reads would see stale data)
  template <int J>
  __device__ __forceinline__ void read_bias() {
    const unsigned ab = a_bias + (unsigned)((((J < 2 ? H + h : h) * 64) + (J & 1) * 32) * 4);
    dsr128f<0>(bb[0], ab);
    dsr128f<32>(bb[1], ab);
    dsr128f<64>(bb[2], ab);
    dsr128f<96>(bb[3], ab);
  }
  template <int S8>
  static __device__ __forceinline__ f16x8 cvt8n(const f32x16& a) {
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)a[S8 + e];
    return o;
  }
  // one slice (0..5) of the epilogue of tile J, accumulator `a`: spread over consecutive steps
  template <int J, int SL>
  __device__ __forceinline__ void epi_slice(const f32x16& a, f16x8& v0, f16x8& v1) {
    constexpr int t = J & 1;
    if constexpr (LEAN && tile_has_bias(J)) {
      if constexpr (SL == 0) v0 = cvt8<0>(a, bb[0], bb[1]);
      if constexpr (SL == 1) v1 = cvt8<8>(a, bb[2], bb[3]);
    } else {
      if constexpr (SL == 0) v0 = cvt8n<0>(a);
      if constexpr (SL == 1) v1 = cvt8n<8>(a);
    }
    if constexpr (J < 2) {  // K tile: lane (key row, hi) holds d = 32t + 8rq + 4hi + i -> k-slot chunks 4t + hi, 4t + 2 + hi
      if constexpr (SL == 2) asm volatile("ds_write_b128 %0, %1" ::"v"(aK[0] ^ (unsigned)(t << 6)), "v"(v0) : "memory");
      if constexpr (SL == 5) asm volatile("ds_write_b128 %0, %1" ::"v"(aK[1] ^ (unsigned)(t << 6)), "v"(v1) : "memory");
    } else if constexpr (J < 4) {  // V tile (swapped operands): lane (d = 32t + r, hi) holds this wave's 32 keys
      if constexpr (SL == 2) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(aV[0]), "v"(v0), "n"(t * 8192) : "memory");
      if constexpr (SL == 5) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(aV[1]), "v"(v1), "n"(t * 8192) : "memory");
    } else {  // Q tile: stays in registers
      if constexpr (SL == 2) qf[2 * t] = v0;
      if constexpr (SL == 5) qf[2 * t + 1] = v1;
    }
  }

  template <int G>
  __device__ __forceinline__ void step(f16x8& e0, f16x8& e1) {
    if constexpr (G < NIT) {
      constexpr int J = G / KS, IT = G % KS;
      constexpr bool SWAP = (J == 2 || J == 3);
      wait_lgkm<younger(G)>();
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 cur = q[G % PF];
      f32x16& acc = (J & 1) ? accB : accA;
      if constexpr (IT == 0) {
        if constexpr (SWAP) {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(xf[0]), "v"(cur));
        } else if constexpr (LEAN) {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(cur), "v"(xf[0]));
        } else {
          f32x16 bv;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            bv[rq * 4 + 0] = bb[rq].x; bv[rq * 4 + 1] = bb[rq].y; bv[rq * 4 + 2] = bb[rq].z; bv[rq * 4 + 3] = bb[rq].w;
          }
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(cur), "v"(xf[0]), "v"(bv));
        }
      } else {
        if constexpr (SWAP) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(xf[IT]), "v"(cur));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(cur), "v"(xf[IT]));
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (IT == SYNC) {
        // tile J + 1 is complete (own DMA pieces landed — the 8 youngest belong to tile J + 2 — then everybody's);
        // this tile's reads are all issued
        unsigned long long tA = 0;
        if constexpr (TM) tA = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (TM) t_sync += __builtin_amdgcn_s_memtime() - tA;
        if constexpr (!LEAN && J + 1 < NT && tile_has_bias(J + 1)) read_bias<(J + 1 < NT ? J + 1 : 0)>();
        if constexpr (LEAN && J == NT - 1) read_bias<NT - 1>();  // q1's own bias, for the epilogue behind the stream
      }
      if constexpr (LEAN && IT == SCH::BIAS_LEAN && J >= 1 && tile_has_bias(J >= 1 ? J - 1 : 0)) read_bias<(J >= 1 ? J - 1 : 0)>();
      if constexpr (G + PF < NIT) read_item<G + PF>();
      // DMA of tile J + 2 into stage (J + 2) % 3 (the one tile J - 1 was read from): pieces at local steps 1-4, 6-9
      constexpr unsigned st = (unsigned)((J + 2) % 3) * TILE_STAGE;
      if constexpr (IT == 0) dma_set_m0(lds_w + st);
      if constexpr (IT == 5) dma_set_m0(lds_w + st + 4096);
      if constexpr (IT >= 1 && IT <= 4) dma_lin<(IT - 1) * 1024>(voff, gimg + (size_t)(J + 2) * TILE_STAGE);
      if constexpr (IT >= 6 && IT <= 9) dma_lin<(IT - 6) * 1024>(voff, gimg + (size_t)(J + 2) * TILE_STAGE + 4096);
      // previous tile's epilogue in this tile's MFMA shadow
      if constexpr (J >= 1 && IT >= EPI0 && IT < EPI0 + 6) epi_slice<J - 1, IT - EPI0>((J & 1) ? accA : accB, e0, e1);
      __builtin_amdgcn_sched_barrier(0);
      step<G + 1>(e0, e1);
    }
  }
  // head prologue: bias of tile 0, then the first PF fragments (the stage was certified by the previous barrier)
  __device__ __forceinline__ void run() {
    if constexpr (!LEAN) read_bias<0>();
    read_item<0>(); read_item<1>(); read_item<2>(); read_item<3>(); read_item<4>(); read_item<5>();
    f16x8 e0, e1;
    step<0>(e0, e1);
    // q1's epilogue (exposed: the attention core needs the fragments now).  LEAN: its bias was read at q1's barrier
    // step, AFTER the last fragments — the last step's counted wait leaves those four reads in flight (found by the CPU
    // replay of the schedule, tests/cpu_sched_check.cpp)
    if constexpr (LEAN) wait_lgkm<0>();
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(accB));  // (tied to the accumulator: see AttnCore::run)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LEAN) {
      qf[2] = cvt8<0>(accB, bb[0], bb[1]);
      qf[3] = cvt8<8>(accB, bb[2], bb[3]);
    } else {
      qf[2] = cvt8n<0>(accB);
      qf[3] = cvt8n<8>(accB);
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Building blocks of the STACK kernel (kernels_stack.hip): the out-projection accumulators (240 AGPRs) stay alive
// through the attention heads — they carry a layout's rows from layer to layer — so the attention core may use arch
// VGPRs only and the out-projection of a head runs right behind its core.
//
// AttnCoreV: the single-tile attention core — S^T = K Q^T (16 MFMAs), in-register softmax, O^T = V^T P^T (16 MFMAs) as ONE
// hand-issued LDS-read queue over the 16 K and 16 V^T fragments (the first V^T reads land under the softmax), cross-half
// reductions through v_permlane32_swap, 1/sum through v_rcp_f32, packed f32 scale-and-shift — with every tile in arch VGPRs
// (r02 before: 4 400 cycles per head, compiler-scheduled; an earlier variant with the output tiles in AGPRs: 2 970).  The probabilities are cast to their fp16 MFMA fragments key tile
// by key tile right behind the exponentials (the 64 score registers shrink to 32 fragment registers before the two
// output tiles come alive), which also puts every VALU write of an MFMA operand many instructions ahead of its use.
// (r02 negative result: running the next key tile's exponentials in the shadow of the O^T MFMAs, a quarter per step,
//  made the core 9 % LONGER — 3 242 vs 2 973 cycles per head, profiles/r02_call38_*; so did issuing the next head's
//  second tile three steps behind SlabPair's barrier instead of at it: +150 cycles per head.)
struct AttnCoreV {
  static constexpr int PF = 4;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f16x8 q[PF];
  unsigned aKr;        // Ks + r*128 + ((hi ^ ksw) << 4): k16-step ks by XOR (ks << 5), key tile kt by offset kt * 4 KiB
  unsigned aVr;        // Vs + r*256 + ((hi ^ (r & 15)) << 4): key chunk (4kt + 2hf) by XOR, d tile by offset 8 KiB
  const f16x8* qf;     // [4] Q fragments (B operand of S^T)
  f32x16 sc[4], o[2];
  f16x8 pfr[8];        // P fragments: pair c = 2kt + hf
  float scale_log2e;
  int S, hi;

  template <int I>  // I in [0, 32): 16 K fragments (kt = I & 3, ks = I >> 2), then 16 V^T fragments (dt, hf, kt)
  __device__ __forceinline__ void read_item() {
    if constexpr (I < 16) {
      dsr128<(I & 3) * 4096>(q[I % PF], aKr ^ (unsigned)((I >> 2) << 5));
    } else if constexpr (I < 32) {
      constexpr int J = I - 16, dt = J & 1, hf = (J >> 1) & 1, kt = J >> 2;
      dsr128<dt * 8192>(q[I % PF], aVr ^ (unsigned)((kt * 4 + hf * 2) << 4));
    }
  }
  template <int I>
  __device__ __forceinline__ void qk_step() {
    if constexpr (I < 16) {
      constexpr int kt = I & 3, ks = I >> 2;
      wait_lgkm<PF - 1>();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(sc[kt]) : "v"(q[I % PF]), "v"(qf[0]));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sc[kt]) : "v"(q[I % PF]), "v"(qf[ks]));
      __builtin_amdgcn_sched_barrier(0);
      read_item<I + PF>();
      qk_step<I + 1>();
    }
  }
  template <int J>
  __device__ __forceinline__ void pv_step() {
    if constexpr (J < 16) {
      constexpr int dt = J & 1, c = J >> 1;
      wait_lgkm<(15 - J < PF - 1 ? 15 - J : PF - 1)>();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (J < 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(o[dt]) : "v"(q[(J + 16) % PF]), "v"(pfr[c]));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(o[dt]) : "v"(q[(J + 16) % PF]), "v"(pfr[c]));
      __builtin_amdgcn_sched_barrier(0);
      read_item<J + 16 + PF>();
      pv_step<J + 1>();
    }
  }
  __device__ __forceinline__ void run(f16x8 (&nf)[4]) {
    read_item<0>(); read_item<1>(); read_item<2>(); read_item<3>();
    qk_step<0>();
    asm volatile("s_nop 15" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]));  // MFMA results -> VALU reads
    __builtin_amdgcn_sched_barrier(0);
    if (S == 125) {
      if (hi) { sc[3][13] = -INFINITY; sc[3][14] = -INFINITY; sc[3][15] = -INFINITY; }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (96 + (i & 3) + 8 * (i >> 2) + 4 * hi >= S) sc[3][i] = -INFINITY;
    }
    float mx = sc[0][0];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sc[kt][i]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const f32x2 sc2 = {scale_log2e, scale_log2e};
    const float nm = -mx * scale_log2e;
    const f32x2 nm2 = {nm, nm};
    f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f32x2 v = {sc[kt][i], sc[kt][i + 1]};
        v = __builtin_elementwise_fma(v, sc2, nm2);
        v.x = __builtin_amdgcn_exp2f(v.x);
        v.y = __builtin_amdgcn_exp2f(v.y);
        sum2 += v;
        pfr[2 * kt + (i >> 3)][i & 7] = (_Float16)v.x;
        pfr[2 * kt + (i >> 3)][(i & 7) + 1] = (_Float16)v.y;
      }
      asm volatile("" : "+v"(pfr[2 * kt]), "+v"(pfr[2 * kt + 1]));  // pin: the scores of this key tile die here
      __builtin_amdgcn_sched_barrier(0);
    }
    float sum = sum2.x + sum2.y;
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
      sum = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = __builtin_amdgcn_rcpf(sum);
    pv_step<0>();
    asm volatile("s_nop 15" : "+v"(o[0]), "+v"(o[1]));
    __builtin_amdgcn_sched_barrier(0);
    const f32x2 inv2 = {inv, inv};
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          f32x2 v = {o[dt][s2 * 8 + e], o[dt][s2 * 8 + e + 1]};
          v = v * inv2;
          nf[dt * 2 + s2][e] = (_Float16)v.x;
          nf[dt * 2 + s2][e + 1] = (_Float16)v.y;
        }
  }
};

// SlabPair: the two out-projection K-slabs of ONE head (d halves 0 / 1; ring stages 0 / 1 = slots 6 / 7 of the head's
// 9-slot cycle) as one 60-item pipeline on the persistent accumulator tiles, directly behind the head's attention core.
// B operands: the head's four normalised output fragments nf[2dt + sx].  While slab 1 runs, tile 0 of the NEXT head
// goes into stage 0 (slab 0's reads were all issued before slab 0's barrier); at slab 1's barrier step that tile and
// slab 1's reads are certified and tile 1 of the next head follows into stage 1 — so the next HeadStream starts with
// its first tile resident and its second in flight, exactly the state its vmcnt(8) protocol expects.
template <int NT2, bool TM = false>
struct SlabPair {
  static constexpr int NIT = 2 * NT2, PF = 6, SYNC = NIT - PF;
  f16x8 q[PF];
  unsigned aS[2];       // stage 0 addresses of the two k16-steps (stage 1: + offset)
  f32x16* acc;
  const f16x8* nf;      // [4]
  const char* gnext;    // image of the next head's tile 0 + wave * 8 KiB (uniform)
  unsigned lds_w;       // lds0 + wave * 8 KiB
  unsigned voff;
  unsigned long long t_sync = 0;

  template <int G>
  __device__ __forceinline__ void read_item() {
    constexpr int C = G / NIT, IT = G % NIT, sx = IT / NT2, t = IT % NT2;
    dsr128<t * 2048 + C * TILE_STAGE>(q[G % PF], aS[sx]);
  }
  template <int G>
  __device__ __forceinline__ void step() {
    if constexpr (G < 2 * NIT) {
      constexpr int C = G / NIT, IT = G % NIT, sx = IT / NT2, t = IT % NT2;
      constexpr int left = 2 * NIT - 1 - G;
      wait_lgkm<(left < PF - 1 ? left : PF - 1)>();
      __builtin_amdgcn_sched_barrier(0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[G % PF], nf[2 * C + sx], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (IT == SYNC) {
        unsigned long long tA = 0;
        if constexpr (TM) tA = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // C = 0: slab 1 has landed; C = 1: the next head's tile 0
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (TM) t_sync += __builtin_amdgcn_s_memtime() - tA;
        if constexpr (C == 1) {  // every read of slab 1 is issued: its stage takes the next head's tile 1
          dma_lin4(voff, gnext + TILE_STAGE, lds_w + TILE_STAGE);
          dma_lin4(voff, gnext + TILE_STAGE + 4096, lds_w + TILE_STAGE + 4096);
        }
      }
      if constexpr (G + PF < 2 * NIT) read_item<G + PF>();
      if constexpr (C == 1) {  // next head's tile 0 -> stage 0
        if constexpr (IT == 0) dma_set_m0(lds_w);
        if constexpr (IT == 5) dma_set_m0(lds_w + 4096);
        if constexpr (IT >= 1 && IT <= 4) dma_lin<(IT - 1) * 1024>(voff, gnext);
        if constexpr (IT >= 6 && IT <= 9) dma_lin<(IT - 6) * 1024>(voff, gnext + 4096);
      }
      __builtin_amdgcn_sched_barrier(0);
      step<G + 1>();
    }
  }
  __device__ __forceinline__ void run() {
    read_item<0>(); read_item<1>(); read_item<2>(); read_item<3>(); read_item<4>(); read_item<5>();
    step<0>();
  }
};

}  // namespace ldm
