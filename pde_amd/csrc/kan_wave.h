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

// Sums of FOUR values over the 64 lanes for the price of little more than one: a swap of half-waves adds the two
// halves of a and of b at once (a's sum ends up in lanes 0..31, b's in lanes 32..63), likewise c and d, a swap of
// rows does the same with the two results, and the sums inside the rows of 16 lanes then finish all four together:
// row 0 holds a's total, row 1 c's, row 2 b's, row 3 d's.  Each total is read back with v_readlane, so the
// results are wave-uniform (scalar registers).
__device__ __forceinline__ void wave_sum4(float& a, float& b, float& c, float& d) {
  const auto ab = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const auto cd = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(d), false, false);
  const float sab = __uint_as_float(ab[0]) + __uint_as_float(ab[1]);
  const float scd = __uint_as_float(cd[0]) + __uint_as_float(cd[1]);
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(sab), __float_as_uint(scd), false, false);
  float v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  const int vi = (int)__float_as_uint(v);
  a = __uint_as_float((unsigned)__builtin_amdgcn_readlane(vi, 0));
  c = __uint_as_float((unsigned)__builtin_amdgcn_readlane(vi, 16));
  b = __uint_as_float((unsigned)__builtin_amdgcn_readlane(vi, 32));
  d = __uint_as_float((unsigned)__builtin_amdgcn_readlane(vi, 48));
}

// m = max(m, |x|) in one instruction (fmaxf(m, fabsf(x)) costs a second one that quiets NaNs)
__device__ __forceinline__ void absmax_into(float& m, float x) {
  asm("v_max_f32 %0, %0, |%1|" : "+v"(m) : "v"(x));
}

// Per-jet maxima for the scales of the fp16 matrix kernels (kan_f16.hip): this lane's maxima of |value| per jet
// slot -> maximum over the wave -> one integer atomic max per slot into out[0..3].  Non-negative floats order like
// their bit patterns, and a maximum does not depend on the order of the atomics: results stay reproducible.
__device__ __forceinline__ void publish_jet_absmax(float m0, float m1, float m2, float m3, unsigned* out) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    m0 = fmaxf(m0, __shfl_xor(m0, d));
    m1 = fmaxf(m1, __shfl_xor(m1, d));
    m2 = fmaxf(m2, __shfl_xor(m2, d));
    m3 = fmaxf(m3, __shfl_xor(m3, d));
  }
  if ((threadIdx.x & 63) == 0) {
    // Thousands of waves combine into the same four words, and atomics on one address are served one after the
    // other: look first (an L2 read; a stale, smaller value only costs an atomic that changes nothing) and add to
    // the queue only when this wave would raise the maximum.  After the first few waves almost none does.
    const float m[4] = {m0, m1, m2, m3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned mine = __float_as_uint(m[j]);
      if (mine > __hip_atomic_load(out + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out + j, mine);
    }
  }
}

// the same for any number of slots (12-jet tensors: value, 6 first and 5 second derivatives)
template <int J>
__device__ __forceinline__ void publish_slot_absmax(float (&m)[J], unsigned* out) {
#pragma unroll
  for (int j = 0; j < J; ++j) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m[j] = fmaxf(m[j], __shfl_xor(m[j], d));
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const unsigned mine = __float_as_uint(m[j]);
      if (mine > __hip_atomic_load(out + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out + j, mine);
    }
  }
}

}  // namespace pdekan
