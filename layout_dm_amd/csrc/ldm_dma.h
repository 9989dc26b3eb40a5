// Device-side helpers shared by the row-stationary kernels (kernels_rowgemm.hip, kernels_fusedattn.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ldm_kernels.h"

namespace ldm {

using dma_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int kLnDp = 512;  // LDS parameter image: multiplier at [0, kLnDp), shift at [kLnDp, 2*kLnDp)

// Linear LDS-DMA for weight images that are stored in global memory EXACTLY as their LDS image (tile
// order, bank swizzle pre-applied on the host): one instruction copies 1 KiB, the per-lane offset is
// lane*16 for every instruction and the 13-bit immediate advances the global AND the LDS address, so
// four consecutive 1-KiB pieces need one M0 write + one SGPR base — 1 issue slot per KiB instead of ~7
// (address VALU, M0, hazard nop, branch).  hipcc does not count these in its vmcnt bookkeeping: every
// wait on them is an explicit s_waitcnt in the kernels.
__device__ __forceinline__ void dma_set_m0(unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0" ::"s"(lds_addr) : "memory");
}
template <int OFF>
__device__ __forceinline__ void dma_lin(unsigned voff, const char* sbase) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// 4 KiB (one M0/base group) at once — prologue use
__device__ __forceinline__ void dma_lin4(unsigned voff, const char* sbase, unsigned lds_addr) {
  asm volatile(
      "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1\n\t"
      "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
      "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
      "global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(voff),
      "s"(sbase), "s"(lds_addr)
      : "memory");
}

// LN-on-load (deferred normalisation) of one token row into register-resident fp16 MFMA fragments:
// y = (x - mean) * rstd * mult + shift with sp[0..) = mult, sp[kLnDp..) = shift staged in LDS; the raw row loads are issued in batches of HB k-steps (2*HB 16-byte loads in flight per lane) and a
// scheduling fence between the batch's loads and its arithmetic: hipcc otherwise sinks every load next to its
// first use (8-14 loads in flight => 5-7 dependent HBM round trips for the 58 loads of a row; see
// profiles/r02_ffn_prologue_epilogue.txt).  2*HB*4 + 4*KS registers live at the peak.
template <int KS, int HB>
__device__ __forceinline__ void load_xf_ln_batched(dma_f16x8 (&xf)[KS], const LnLoad& ln, int m, int hi, const float* sp) {
  const float2 st = ln.stats[m];
  const float* xr = ln.x + (size_t)m * ln.ldx + hi * 8;
  const float* mp = sp + hi * 8;
#pragma unroll
  for (int b0 = 0; b0 < KS; b0 += HB) {
    float4 raw[HB][2];
#pragma unroll
    for (int i = 0; i < HB; ++i)
      if (b0 + i < KS) {
        raw[i][0] = *reinterpret_cast<const float4*>(xr + (b0 + i) * 16);
        raw[i][1] = *reinterpret_cast<const float4*>(xr + (b0 + i) * 16 + 4);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < HB; ++i)
      if (b0 + i < KS) {
        const int ks = b0 + i;
        const float4 a = raw[i][0], b = raw[i][1];
        const float4 ga = *reinterpret_cast<const float4*>(mp + ks * 16);
        const float4 gb = *reinterpret_cast<const float4*>(mp + ks * 16 + 4);
        const float4 sa = *reinterpret_cast<const float4*>(mp + kLnDp + ks * 16);
        const float4 sb = *reinterpret_cast<const float4*>(mp + kLnDp + ks * 16 + 4);
        xf[ks][0] = (_Float16)fmaf((a.x - st.x) * st.y, ga.x, sa.x);
        xf[ks][1] = (_Float16)fmaf((a.y - st.x) * st.y, ga.y, sa.y);
        xf[ks][2] = (_Float16)fmaf((a.z - st.x) * st.y, ga.z, sa.z);
        xf[ks][3] = (_Float16)fmaf((a.w - st.x) * st.y, ga.w, sa.w);
        xf[ks][4] = (_Float16)fmaf((b.x - st.x) * st.y, gb.x, sb.x);
        xf[ks][5] = (_Float16)fmaf((b.y - st.x) * st.y, gb.y, sb.y);
        xf[ks][6] = (_Float16)fmaf((b.z - st.x) * st.y, gb.z, sb.z);
        xf[ks][7] = (_Float16)fmaf((b.w - st.x) * st.y, gb.w, sb.w);
        // pin the fragment here (hipcc's IR-level sinking otherwise moves the arithmetic below the next batch's
        // loads: raw rows + parameters of several batches live at once -> scratch) and fence every second k-step
        // (the 4 parameter reads per k-step would otherwise all be hoisted to the top of the batch)
        asm volatile("" : "+v"(xf[ks]));
        if (i & 1) __builtin_amdgcn_sched_barrier(0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Same normalisation, but the row is loaded in ACCUMULATOR layout — lane (row, hi) fetches columns 8g + 4hi .. +3 of
// every 8-column group g — and groups 2ks, 2ks+1 form fragment ks: element e <-> column 16ks + 8(e>>2) + 4hi + (e&3),
// the MFMA k-slot order (ldm_pack::kslot).  For weights whose K axis is packed in that order; it is the layout in
// which the fused kernels hold a row in their accumulators, so fragments built from memory and fragments built from
// registers are interchangeable.  NG = valid 8-column groups (58 for d_model 464), GB = groups per load batch.
template <int KS, int NG, int GB>
__device__ __forceinline__ void load_xf_ln_acc(dma_f16x8 (&xf)[KS], const LnLoad& ln, size_t m, int hi, const float* sp) {
  const float2 st = ln.stats[m];
  const float* xr = ln.x + m * ln.ldx + hi * 4;
  const float* mp = sp + hi * 4;
#pragma unroll
  for (int g0 = 0; g0 < NG; g0 += GB) {
    float4 raw[GB];
#pragma unroll
    for (int i = 0; i < GB; ++i)
      if (g0 + i < NG) raw[i] = *reinterpret_cast<const float4*>(xr + (g0 + i) * 8);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const int gg = g0 + i;
      if (gg < NG) {
        const float4 a = raw[i];
        const float4 ga = *reinterpret_cast<const float4*>(mp + gg * 8);
        const float4 sa = *reinterpret_cast<const float4*>(mp + kLnDp + gg * 8);
        const int ks = gg >> 1, e0 = (gg & 1) * 4;
        xf[ks][e0 + 0] = (_Float16)fmaf((a.x - st.x) * st.y, ga.x, sa.x);
        xf[ks][e0 + 1] = (_Float16)fmaf((a.y - st.x) * st.y, ga.y, sa.y);
        xf[ks][e0 + 2] = (_Float16)fmaf((a.z - st.x) * st.y, ga.z, sa.z);
        xf[ks][e0 + 3] = (_Float16)fmaf((a.w - st.x) * st.y, ga.w, sa.w);
        if (gg & 1) asm volatile("" : "+v"(xf[ks]));
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace ldm
