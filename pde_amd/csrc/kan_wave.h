// Wavefront-level sums for gfx950 that stay on the VALU: DPP lane permutes inside
// a row of 16 lanes, v_permlane16_swap / v_permlane32_swap across rows.  The
// generic shuffles go through the LDS crossbar (ds_bpermute) and cost an LDS
// round trip per step, which a dependent reduction chain cannot hide.
#pragma once
#include <hip/hip_runtime.h>

namespace pdekan {

// v + (the same value 32 / 16 lanes away), in every lane
__device__ __forceinline__ float sum_xor32(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor16(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int kCtrl>
__device__ __forceinline__ float dpp_move(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), kCtrl, 0xf, 0xf, false));
}

// sum over the 64 lanes, result in every lane.  Steps: neighbour in the quad (quad_perm 1,0,3,2),
// other pair of the quad (quad_perm 2,3,0,1), other quad of the 8 lanes (row_half_mirror), other
// half of the row (row_mirror), other row of the pair, other half of the wave.
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  v = sum_xor16(v);
  return sum_xor32(v);
}

}  // namespace pdekan
