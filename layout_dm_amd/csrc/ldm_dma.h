// LDS-DMA helpers of the stack kernel (kernels_stack.hip, ldm_pipes.h).
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

}  // namespace ldm
