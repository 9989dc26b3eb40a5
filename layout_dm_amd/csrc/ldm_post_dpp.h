// Lane-group policies of ldm_post_token.h for gfx950: reductions WITHOUT the LDS.  A __shfl_xor is a ds_bpermute_b32 (an
// LDS round trip, ~130 cycles, and the steps of a reduction are a dependent chain); here a butterfly runs on DPP moves
// inside a 16-lane row (quad_perm, row_half_mirror, row_mirror: one VALU issue each) and, for the 64-lane group, the
// two cross-row exchanges go through gfx950's v_permlane16_swap / v_permlane32_swap.
//
//   DppGroup<16>   one DPP row per token: four tokens per wavefront.  Row operations never cross a row, so the groups of
//                  a wavefront may diverge from each other (whole rows active or inactive).
//   DppGroup<64>   one wavefront per token.
#pragma once
#include <hip/hip_runtime.h>

#include "ldm_post_token.h"

namespace ldm_post {

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_movd(double v) {
  return __hiloint2double(dpp_mov<CTRL>(__double2hiint(v)), dpp_mov<CTRL>(__double2loint(v)));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_movd_rows(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

__device__ __forceinline__ float swap16_f(float x, float& other) {
  const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  other = __uint_as_float(s[1]);
  return __uint_as_float(s[0]);
}
__device__ __forceinline__ float swap32_f(float x, float& other) {
  const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  other = __uint_as_float(s[1]);
  return __uint_as_float(s[0]);
}
__device__ __forceinline__ int swap16_i(int x, int& other) {
  const auto s = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
  other = (int)s[1];
  return (int)s[0];
}
__device__ __forceinline__ int swap32_i(int x, int& other) {
  const auto s = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
  other = (int)s[1];
  return (int)s[0];
}

// FAST: exp / log on the hardware's v_exp_f32 / v_log_f32 (the fast numerics mode; ~1e-6 relative) instead of libm
template <int NL_, bool FAST = false>
struct DppGroup {
  static_assert(NL_ == 16 || NL_ == 64, "one DPP row or one wavefront");
  static constexpr int NL = NL_;
  int l;  // lane inside the group
  __device__ __forceinline__ int lane() const { return l; }
  __device__ __forceinline__ float exp(float x) const { return FAST ? __expf(x) : expf(x); }
  __device__ __forceinline__ float log(float x) const { return FAST ? __logf(x) : logf(x); }

  __device__ __forceinline__ float gmax(float v) const {
    v = fmaxf(v, __int_as_float(dpp_mov<kDppXor1>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_mov<kDppXor2>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_mov<kDppHalfMirror>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_mov<kDppMirror>(__float_as_int(v))));
    if constexpr (NL_ == 64) {
      float o;
      v = swap16_f(v, o); v = fmaxf(v, o);
      v = swap32_f(v, o); v = fmaxf(v, o);
    }
    return v;
  }
  __device__ __forceinline__ float gsum(float v) const {
    v += __int_as_float(dpp_mov<kDppXor1>(__float_as_int(v)));
    v += __int_as_float(dpp_mov<kDppXor2>(__float_as_int(v)));
    v += __int_as_float(dpp_mov<kDppHalfMirror>(__float_as_int(v)));
    v += __int_as_float(dpp_mov<kDppMirror>(__float_as_int(v)));
    if constexpr (NL_ == 64) {
      float o;
      v = swap16_f(v, o); v += o;
      v = swap32_f(v, o); v += o;
    }
    return v;
  }
  __device__ __forceinline__ int gsumi(int v) const {
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppHalfMirror>(v);
    v += dpp_mov<kDppMirror>(v);
    if constexpr (NL_ == 64) {
      int o;
      v = swap16_i(v, o); v += o;
      v = swap32_i(v, o); v += o;
    }
    return v;
  }
  __device__ __forceinline__ double gsumd(double v) const {
    v += dpp_movd<kDppXor1>(v);
    v += dpp_movd<kDppXor2>(v);
    v += dpp_movd<kDppHalfMirror>(v);
    v += dpp_movd<kDppMirror>(v);
    if constexpr (NL_ == 64) {
      int olo, ohi;
      int lo = swap16_i(__double2loint(v), olo), hi = swap16_i(__double2hiint(v), ohi);
      v = __hiloint2double(hi, lo) + __hiloint2double(ohi, olo);
      lo = swap32_i(__double2loint(v), olo); hi = swap32_i(__double2hiint(v), ohi);
      v = __hiloint2double(hi, lo) + __hiloint2double(ohi, olo);
    }
    return v;
  }
  // (value, class) -> the largest value, the smallest class among equals — on every lane of the group
  __device__ __forceinline__ void gargmax(float& bv, int& bi) const {
    auto take = [&](float ov, int oi) {
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    };
    take(__int_as_float(dpp_mov<kDppXor1>(__float_as_int(bv))), dpp_mov<kDppXor1>(bi));
    take(__int_as_float(dpp_mov<kDppXor2>(__float_as_int(bv))), dpp_mov<kDppXor2>(bi));
    take(__int_as_float(dpp_mov<kDppHalfMirror>(__float_as_int(bv))), dpp_mov<kDppHalfMirror>(bi));
    take(__int_as_float(dpp_mov<kDppMirror>(__float_as_int(bv))), dpp_mov<kDppMirror>(bi));
    if constexpr (NL_ == 64) {
      float ov; int oi;
      bv = swap16_f(bv, ov); bi = swap16_i(bi, oi); take(ov, oi);
      bv = swap32_f(bv, ov); bi = swap32_i(bi, oi); take(ov, oi);
    }
  }
  // inclusive prefix sum over the lanes of the group: Hillis-Steele inside a row (row_shr 1 / 2 / 4 / 8, zeros shifted
  // in); for the wavefront group the totals of the preceding rows through row_bcast:15 (rows 1, 3) and row_bcast:31
  // (rows 2, 3)
  __device__ __forceinline__ double gscan(double v) const {
    v += dpp_movd_rows<0x111, 0xF>(v);
    v += dpp_movd_rows<0x112, 0xF>(v);
    v += dpp_movd_rows<0x114, 0xF>(v);
    v += dpp_movd_rows<0x118, 0xF>(v);
    if constexpr (NL_ == 64) {
      v += dpp_movd_rows<0x142, 0xA>(v);
      v += dpp_movd_rows<0x143, 0xC>(v);
    }
    return v;
  }
  // the group's scratch lives in LDS and is written and read by lanes of ONE wavefront: LDS operations of a wavefront
  // complete in order, so draining the counter is all the ordering there is to do
  __device__ __forceinline__ void sync() const {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
  }
};

}  // namespace ldm_post
