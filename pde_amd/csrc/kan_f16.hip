// Hidden KAN layers on the fp16 matrix cores of gfx950 with fp32 accuracy, J = 4
// jets per scalar (value, d/dS, d/dt, d2/dS2): forward, input-jet gradients and
// weight gradients of layers whose knot rows are equally spaced or nearly so.
//
// Reference computation: KANLinear.forward (efficient_kan/kan.py:153-166) and its
// autograd backward, on jets (the derivatives run1.py:121-124 take with jacrev /
// hessian).
//
// Split operands.  Every fp32 operand x of the three contractions
//   forward        Y [(p,j)][o]  = sum_k E[(p,j)][k] * W[o][k]
//   input grads    gE[(p,j)][k]  = sum_o gY[(p,j)][o] * W[o][k]     (+ chain rule through the basis)
//   weight grads   gW[k][o]      = sum_(p,j) E[(p,j)][k] * gY[(p,j)][o]
// (k = (input feature, column): the 4 live cubic bases of the feature + its base-activation column, the other
// columns of the feature zero) is first multiplied by a power of two that brings its tensor's largest magnitude
// just under 2^13 and then cut into two fp16 pieces, hi = fp16(x) (round to nearest) and lo = fp16(x - hi) (the
// subtraction is exact in fp32).  hi + lo carries 22 significant bits of x wherever |x| is within 2^-16 of the
// tensor's maximum, and an absolute error below 2^-38 of that maximum elsewhere (fp16 subnormals).  A product is
//     hi*hi + hi*lo + lo*hi
// three v_mfma_f32_16x16x32_f16 whose fp16 x fp16 products are exact in the fp32 accumulator; the term left out
// is below 2^-22 |a b|.  The powers of two are divided out of the fp32 result (exact).  Per-jet scales: the jets
// of a tensor differ by orders of magnitude (a second derivative against a value), so every jet slot j has its
// own scale wherever the slot is a row index of the contraction; where it is summed over (weight gradients) the
// product of the two operands' scales is kept the same for every j.
// The maxima come from the kernels that produced the tensors (an epilogue of the producing kernel, or
// jet_absmax_kernel), as float bit patterns combined with integer atomic max: the result does not depend on
// the order of the atomics, so everything stays bitwise reproducible.
//
// Matrix work per fp32-accurate product: 3/16 of the fp32 matrix time (the fp32 MFMA runs at 1/16 of the fp16
// rate), against 6/16 for the three-piece bf16 cut, and dense over the G+4 columns of a feature where the
// vector-ALU kernels of kan_sparse.hip do 5: at grid 10 that is 3 * 14 / (16 * 5) = 0.53 of their ALU time.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kan_math.h"
#include "kan_launch.h"
#include <type_traits>

namespace pdekan {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kF16Target = 13;               // scaled maxima sit in [2^12, 2^13): 8x headroom under fp16's 65504
constexpr int kKnotPadU = 28;                // staged knot row, equally spaced: G+7 knots + 1/h, G <= 20
constexpr int kKnotPadN = 36;                // nearly equally spaced: 3 + 4 repeated end knots around the G+7 knots
// the same for the kernels that stage the rows of the whole layer (a grid-10 row of either kind fits 28 words: the two
// variants of a kernel then ask for the same LDS and can share a launch without costing the equally spaced one a
// resident block)
constexpr int layer_knot_pad(int nk, bool gen) { return gen ? (nk + 8 <= kKnotPadU ? kKnotPadU : kKnotPadN) : kKnotPadU; }

// ---- scales ---------------------------------------------------------------------------------------------
// 2^k with bound * 2^k in [2^(target-1), 2^target); bound = 0 or denormal gives the largest scale allowed
__device__ __forceinline__ float pow2_scale(float bound, int target = kF16Target) {
  const int e = (int)((__float_as_uint(bound) >> 23) & 255u) - 127;
  int k = target - 1 - e;
  k = k > 96 ? 96 : (k < -96 ? -96 : k);
  return __uint_as_float((unsigned)(k + 127) << 23);
}
__device__ __forceinline__ float rcp_pow2(float s) {      // exact reciprocal of a power of two
  return __uint_as_float((254u << 23) - __float_as_uint(s));
}

// Upper bounds of the expanded jets E[(p, j)][k] of a layer from the per-jet maxima of its input jets:
//   E0 = B or act(a0);  E1 = phi' a1;  E2 = phi' a2;  E3 = phi'' a1^2 + phi' a3,   |B'| <= 1/h, |B''| <= 2/h^2,
//   |silu'| <= 1.1, |silu''| <= 0.5 (identity: 1, 0).
struct JetScales { float s[4]; };
__device__ __forceinline__ JetScales expanded_jet_scales(const unsigned* amax, float inv_h_max) {
  const float a0 = __uint_as_float(amax[0]), a1 = __uint_as_float(amax[1]), a2 = __uint_as_float(amax[2]),
              a3 = __uint_as_float(amax[3]);
  const float c1 = fmaxf(1.1f, 1.5f * inv_h_max), c2 = fmaxf(0.5f, 3.0f * inv_h_max * inv_h_max);
  JetScales r;
  r.s[0] = pow2_scale(fmaxf(1.0f, a0));
  r.s[1] = pow2_scale(c1 * a1);
  r.s[2] = pow2_scale(c1 * a2);
  r.s[3] = pow2_scale(c2 * a1 * a1 + c1 * a3);
  return r;
}

// largest |1/h| over the layer's features (ginfo holds 1/h or -1/h), the same in every lane
__device__ __forceinline__ float max_inv_h(const float* ginfo, int I, int lane) {
  float m = 0.0f;
  for (int i = lane; i < I; i += 64) m = fmaxf(m, fabsf(ginfo[i]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  return m;
}

// out[j] = max(out[j], bits of max |t[e][j]|) over e < n4: per-jet maxima of a jet tensor laid out [...][4]
__global__ void jet_absmax_kernel(const f32x4* __restrict__ t, int64_t n4, unsigned* __restrict__ out) {
  f32x4 m = {0.f, 0.f, 0.f, 0.f};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; e + 3 * stride < n4; e += 4 * stride) {           // 4 loads in flight per thread
    const f32x4 v0 = t[e], v1 = t[e + stride], v2 = t[e + 2 * stride], v3 = t[e + 3 * stride];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      m[j] = fmaxf(fmaxf(m[j], fmaxf(fabsf(v0[j]), fabsf(v1[j]))), fmaxf(fabsf(v2[j]), fabsf(v3[j])));
  }
  for (; e < n4; e += stride) {
    const f32x4 v = t[e];
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], fabsf(v[j]));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m[j] = fmaxf(m[j], __shfl_xor(m[j], d));
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicMax(out + j, __float_as_uint(m[j]));
  }
}

// out[0] = max(out[0], bits of max |w[e]|)
__global__ void absmax_kernel(const float* __restrict__ w, int64_t n, unsigned* __restrict__ out) {
  float m = 0.0f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(w[e]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// per-jet maxima of this lane's values -> wave maximum -> one integer atomic max per jet and wave
__device__ __forceinline__ void publish_jet_max(f32x4 m, unsigned* out, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m[j] = fmaxf(m[j], __shfl_xor(m[j], d));
  }
  if (lane == 0) {                              // look before queueing on the four shared words: see publish_jet_absmax
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned mine = __float_as_uint(m[j]);
      if (mine > __hip_atomic_load(out + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out + j, mine);
    }
  }
}

// ---- the two-piece cut ----------------------------------------------------------------------------------
// (x0, x1) -> one dword of each plane, x0 in the low half: hi = fp16(x) to nearest, lo = fp16(x - hi) (fp32
// subtraction inside the mixed-precision fma, rounded once to fp16)
__device__ __forceinline__ void split_pair16(float x0, float x1, unsigned& h, unsigned& l) {
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
  asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(x0));
  asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x1));
}

struct Planes16 { u32x2 h, l; };                           // the 4 jets of one scalar, two planes

__device__ __forceinline__ Planes16 split_jet16(f32x4 v) {
  Planes16 r;
  unsigned h0, l0, h1, l1;
  split_pair16(v[0], v[1], h0, l0);
  split_pair16(v[2], v[3], h1, l1);
  r.h = (u32x2){h0, h1};
  r.l = (u32x2){l0, l1};
  return r;
}

// ---- expansion of one (point, feature) pair ---------------------------------------------------------------
struct FeatGrid { float g0, glast, inv_h; };
struct PairCols { f32x4 col[5]; int c0; };                 // 4 live spline columns c0..c0+3 and the base column

// GEN = false: knots equally spaced (fixed polynomial pieces, interval from the quotient alone); GEN = true:
// nearly so (general triangle on the knots around the quotient guess; the row is staged with repeated end knots)
template <int ACT, bool GEN>
__device__ __forceinline__ void expand_scaled(const float* knots, int nk, const FeatGrid& fg, f32x4 jet,
                                              const JetScales& sc, PairCols& out) {
  Basis bs;
  if constexpr (GEN) eval_basis_near_uniform(knots, nk, fg.g0, fg.glast, fg.inv_h, jet[0], bs);
  else eval_basis_uniform_by_quotient(knots, nk, fg.g0, fg.glast, fg.inv_h, jet[0], bs);
  const Act av = eval_act(ACT, jet[0]);
  const float a1 = jet[1] * sc.s[1], a2 = jet[2] * sc.s[2], a3 = jet[3] * sc.s[3], q3 = jet[1] * jet[1] * sc.s[3];
#pragma unroll
  for (int m = 0; m < 4; ++m)
    out.col[m] = (f32x4){bs.b[m] * sc.s[0], bs.d1[m] * a1, bs.d1[m] * a2, bs.d2[m] * q3 + bs.d1[m] * a3};
  out.col[4] = (f32x4){av.f0 * sc.s[0], av.f1 * a1, av.f1 * a2, av.f2 * q3 + av.f1 * a3};
  out.c0 = bs.c0;
}

// ---- E tile of a wave in LDS: [2 planes][ROWS columns][RW words], a row = one expanded column k holding the
// (p, j) of the wave's points as fp16 (2 words per point) -------------------------------------------------------
// 16 points x 4 jets x fp16 = 128 bytes + 32 bytes (dump slot).  40 words: the 8 consecutive rows that a 32-lane group
// of a transposed read touches (8 words each) start 40 * d mod 64 = {0, 40, 16, 56, 32, 8, 48, 24} banks apart: no conflicts.
constexpr int kRowWords = 40;

// Order of the expanded columns inside a k-step of 32 (the same for the A operand's transposed reads and the packed
// B operand): element e = 0..7 of lane group kb = 0..3 is column 16*(kb>>1) + 8*(e>>2) + 4*(kb&1) + (e&3) of the k-step, so
// that the two lane groups that share an LDS access (kb = 0, 1 or 2, 3) read 8 consecutive rows of the tile.
__host__ __device__ constexpr int kstep_col(int kb, int e) { return 16 * (kb >> 1) + 8 * (e >> 2) + 4 * (kb & 1) + (e & 3); }

// Order of a feature's 16 columns inside a k-step of the 32 x 32 x 16 forward (A operand's transposed reads and packed B operand alike): element
// e of lane half h is column 8h + 2 (e & 3) + (e >> 2).  The first transposed read of a 32-lane half then takes 4
// rows of the E tile two apart and the second the 4 rows between them, so the two 16-lane groups of the half - which
// read different (point, row) blocks of the same rows - land on different banks.
__host__ __device__ constexpr int wide_col(int h, int e) { return 8 * h + 2 * (e & 3) + (e >> 2); }

// park a pair's planes: clear the columns written one stage earlier, write the live ones
template <int C1, int PLANE>
__device__ __forceinline__ void park_pair16(unsigned* base, unsigned* dump, const Planes16 (&pp)[5], int c0, int& prev_c0) {
  constexpr int C = C1 - 1;
  const u32x2 zero = {0u, 0u};
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = prev_c0 + m;
    unsigned* dst = (c >= 0 && c < C) ? base + c * kRowWords : dump;
    *reinterpret_cast<u32x2*>(dst) = zero;
    *reinterpret_cast<u32x2*>(dst + PLANE) = zero;
  }
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const int c = (m < 4) ? c0 + m : C;
    unsigned* dst = (m == 4 || (c >= 0 && c < C)) ? base + c * kRowWords : dump;
    *reinterpret_cast<u32x2*>(dst) = pp[m].h;
    *reinterpret_cast<u32x2*>(dst + PLANE) = pp[m].l;
  }
  prev_c0 = c0;
}

// A operand of one (k-step, m-tile): 8 columns of this lane's (p, j), transposed out of the E tile by the
// hardware (a 16-lane group reads 4 columns x 16 (p, j), every lane gets its (p, j) of the 4 columns)
struct ATile16 { u32x4 h, l; };

__device__ __forceinline__ u32x4 tr_read8(const unsigned* row0, const unsigned* row1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)row0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)row1);
  const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
  return (u32x4){a.x, a.y, b.x, b.y};
}

template <int PLANE>
__device__ __forceinline__ ATile16 read_atile16(const unsigned* sE, int off0, int off1, int m) {
  ATile16 t;
  const unsigned* p0 = sE + off0 + 8 * m;    // 16 (p, j) per m-tile = 8 words
  const unsigned* p1 = sE + off1 + 8 * m;
  t.h = tr_read8(p0, p1);
  t.l = tr_read8(p0 + PLANE, p1 + PLANE);
  return t;
}

// acc += a * b to fp32 accuracy: the two cross terms first, the leading term last
__device__ __forceinline__ f32x4 mfma3(f32x4 c, u32x4 ah, u32x4 al, u32x4 bh, u32x4 bl) {
  const f16x8 Ah = __builtin_bit_cast(f16x8, ah), Al = __builtin_bit_cast(f16x8, al);
  const f16x8 Bh = __builtin_bit_cast(f16x8, bh), Bl = __builtin_bit_cast(f16x8, bl);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh, c, 0, 0, 0);
  return c;
}

// the same for the 4 output tiles of one k-step, product by product: consecutive matrix instructions write different
// accumulators (a dependent one would hold the wave's issue until its predecessor has left the pipe)
__device__ __forceinline__ void mfma3x4(f32x4 (&acc)[4], u32x4 ah, u32x4 al, const u32x4 (&bh)[4], const u32x4 (&bl)[4]) {
  const f16x8 Ah = __builtin_bit_cast(f16x8, ah), Al = __builtin_bit_cast(f16x8, al);
#pragma unroll
  for (int n = 0; n < 4; ++n)
    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, __builtin_bit_cast(f16x8, bh[n]), acc[n], 0, 0, 0);
#pragma unroll
  for (int n = 0; n < 4; ++n)
    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, __builtin_bit_cast(f16x8, bl[n]), acc[n], 0, 0, 0);
#pragma unroll
  for (int n = 0; n < 4; ++n)
    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, __builtin_bit_cast(f16x8, bh[n]), acc[n], 0, 0, 0);
}

#ifndef PDEKAN_SCHED_VARIANT
#define PDEKAN_SCHED_VARIANT 0
#endif
// scheduling hint for a fenced region: CNT matrix instructions, each followed by up to PER other instructions
template <int CNT, int PER>
__device__ __forceinline__ void interleave() {
#if PDEKAN_SCHED_VARIANT == 0 || PDEKAN_SCHED_VARIANT == 3
#pragma unroll
  for (int k = 0; k < CNT; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one MFMA
    __builtin_amdgcn_sched_group_barrier(0x3b2, PDEKAN_SCHED_VARIANT == 3 ? 2 : PER, 0);         // VALU / VMEM / DS
  }
#endif
}
#if PDEKAN_SCHED_VARIANT == 2
#define PDEKAN_REGION_FENCE()
#else
#define PDEKAN_REGION_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// ==========================================================================================================
// W planes, both scaled by the power of two that brings max |w| under 2^13.
//   forward (B operand, K = columns): CP = columns per feature in the K index (C1 rounded up so that 4 features are
//     whole k-steps of 32: 14 -> 16, 24 -> 24), a stage = 4 features = KS = CP/8 k-steps;
//     wplf[O/64][I/4 stages][KS][4 n][2 planes][64 lanes] x 16 bytes,
//     lane (r = lane&15, kb = lane>>4) holds output 64*by + 16n + r, columns 32*ks + kstep_col(kb, 0 .. 7) of the stage
//   input gradients (B operand, K = outputs): the columns g = f * C1 + c of a PAIR of stages (8 features) in tiles of
//     16, no padding: the even stage multiplies the first TA tiles (at grid 10 the last of them reaches 8 columns into
//     the odd stage's first feature), the odd stage the other TB (BwiTiles);
//     wplb[O/64][I/8 pairs][even: 2 ks x TA n | odd: 2 ks x TB n][2 planes][64 lanes] x 16 bytes,
//     lane (r, kb) holds column 16 * tile + r of the pair, outputs 64*by + 32*ks + 8*kb .. + 7
// ==========================================================================================================
template <int C1>
struct BwiTiles {
  static_assert((8 * C1) % 16 == 0, "a pair of stages is whole column tiles");
  static constexpr int TA = (4 * C1 + 15) / 16, TB = (8 * C1) / 16 - TA;     // 14: 4 + 3; 24: 6 + 6
  static constexpr int RING = TA * 16;                     // column slots of a parked gE tile: column g sits at g % RING
  static constexpr int OFFB = 4 * C1;                      // first column of the odd stage
  static constexpr bool WRAP = OFFB % RING != 0;           // the odd stage's columns start inside the even stage's last tile
  static constexpr int NRP = 2 * (TA + TB);                // regions (k-step, column tile) per pair
  static constexpr int kPairVecs = NRP * 2 * 64;           // u32x4 per pair of stages
  // the B tiles come through a ring of 4 registers sets: where in the ring the stages of two pairs start
  static constexpr int PH1 = (2 * TA) % 4, PH2 = NRP % 4, PH3 = (NRP + 2 * TA) % 4;
  static_assert((2 * NRP) % 4 == 0, "two pairs bring the ring back to where it started");
};

// WIDE (C1 = 14): the forward image is laid out for fwd12w_f16_kernel instead,
//   wplf[O/64][I features][2 n][2 planes][64 lanes] x 16 bytes, lane (r = lane&31, h = lane>>5) holds output
//   64*by + 32n + r, columns wide_col(h, 0 .. 7) of the feature (the same number of bytes)
// SPARSE (C1 = 14): the forward image is laid out for fwd_sp_f16_kernel (see there),
//   wsp[O/64][I/4 stages][4 n][2 planes][2 halves][64 lanes] x 16 bytes, lane (oc = lane&15, tb = lane>>4) holds output
//   64*by + 16n + oc; half h, element e: feature tb/2 + 2h of the stage, column 4*(e&3) + 2*(tb&1) + (e>>2)
template <int C1, bool WIDE, bool SPARSE = false>
__global__ void pack_wplanes16_kernel(const float* __restrict__ wp, const unsigned* __restrict__ wmax,
                                      u32x4* __restrict__ wplf, u32x4* __restrict__ wplb, int I, int O) {
  constexpr int CP = (C1 + 7) / 8 * 8, KS = CP / 8;
  using T = BwiTiles<C1>;
  const float sw = pow2_scale(__uint_as_float(wmax[0]));
  const int nst = I / 4, npair = I / 8;
  const int total_f = (O / 64) * nst * KS * 4 * 64;
  const int total_b = (O / 64) * npair * T::NRP * 64;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total_f + total_b; t += gridDim.x * blockDim.x) {
    float v[8];
    u32x4* dst;
    int lo_plane = 64;                      // where the low plane goes, in u32x4 behind the high plane
    if (SPARSE && t < total_f) {
      const int lane = t & 63, half = (t >> 6) & 1, n = (t >> 7) & 3, st = (t >> 9) % nst, by = (t >> 9) / nst;
      const int o = by * 64 + n * 16 + (lane & 15), tb = lane >> 4;
      const int f = st * 4 + (tb >> 1) + 2 * half;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 4 * (e & 3) + 2 * (tb & 1) + (e >> 2);
        v[e] = (c < C1) ? wp[((size_t)o * I + f) * C1 + c] * sw : 0.0f;
      }
      dst = wplf + ((((size_t)(by * nst + st) * 4 + n) * 2) * 2 + half) * 64 + lane;
      lo_plane = 128;
    } else if (WIDE && t < total_f) {
      const int lane = t & 63, n = (t >> 6) & 1, f = (t >> 7) % I, by = (t >> 7) / I;
      const int o = by * 64 + n * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = wide_col(h, e);
        v[e] = (c < C1) ? wp[((size_t)o * I + f) * C1 + c] * sw : 0.0f;
      }
      dst = wplf + ((((size_t)by * I + f) * 2 + n) * 2) * 64 + lane;
    } else if (t < total_f) {
      const int lane = t & 63, n = (t >> 6) & 3;
      const int ks = (t >> 8) % KS, st = ((t >> 8) / KS) % nst, by = (t >> 8) / (KS * nst);
      const int o = by * 64 + n * 16 + (lane & 15), kb = lane >> 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = ks * 32 + kstep_col(kb, e);
        const int f = st * 4 + k / CP, c = k % CP;
        v[e] = (c < C1) ? wp[((size_t)o * I + f) * C1 + c] * sw : 0.0f;
      }
      dst = wplf + ((((size_t)(by * nst + st) * KS + ks) * 4 + n) * 2) * 64 + lane;
    } else {
      const int u = t - total_f;
      const int lane = u & 63, q = (u >> 6) % T::NRP;
      const int pr = ((u >> 6) / T::NRP) % npair, by = (u >> 6) / (T::NRP * npair);
      const bool odd = q >= 2 * T::TA;
      const int q2 = odd ? q - 2 * T::TA : q, nb = odd ? T::TB : T::TA;
      const int ks = q2 / nb, tile = (odd ? T::TA : 0) + q2 % nb;
      const int g = tile * 16 + (lane & 15), kb = lane >> 4;
      const int f = pr * 8 + g / C1, c = g % C1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int o = by * 64 + ks * 32 + kb * 8 + e;
        v[e] = wp[((size_t)o * I + f) * C1 + c] * sw;
      }
      dst = wplb + ((((size_t)(by * npair + pr)) * T::NRP + q) * 2) * 64 + lane;
    }
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair16(v[2 * e], v[2 * e + 1], h[e], l[e]);
    dst[0] = (u32x4){h[0], h[1], h[2], h[3]};
    dst[lo_plane] = (u32x4){l[0], l[1], l[2], l[3]};
  }
}

// ==========================================================================================================
// forward.  M = (point, jet): 16 rows = 4 points, N = 16 outputs, K = 32 columns.  Wave = 16 points x 64 outputs
// (16 accumulator tiles) over all features, a stage of 4 features (KS k-steps, 4*KS regions of 12 matrix
// instructions) at a time; the pairs of the next stage are expanded, cut and parked between the matrix
// instructions of this one.  B operands (W planes) go from L2 straight into registers one stage ahead, A operands
// come out of the wave's E tile by transposed reads two regions ahead.  No block barrier in the loop.
// Runs when every feature of the layer has equally spaced knots (GEN = false) / at least nearly so and not all
// equally (GEN = true); otherwise it returns at once and fwd_sparse_kernel, launched beside it, does the layer.
// ==========================================================================================================
struct WTiles16 { u32x4 h[4], l[4]; };                     // B operands of one k-step: [n]
struct WTile16 { u32x4 h, l; };                            // B operands of one (k-step, output tile)

__device__ __forceinline__ void load_w16(WTiles16& W, const u32x4* __restrict__ src, int lane, int n) {
  const u32x4* p = src + (n * 2) * 64 + lane;
  W.h[n] = p[0];
  W.l[n] = p[64];
}

template <int C1, int ACT, bool GEN>
struct FwdF16 {
  static constexpr int CP = (C1 + 7) / 8 * 8;
  static constexpr int KS = CP / 8;                        // k-steps of 32 columns per stage of 4 features
  static constexpr int NR = 4 * KS;                        // regions (k-step, m-tile) per stage
  static constexpr int ROWS = 4 * C1 + 1;                  // + one row of zeros for the pad columns
  static constexpr int PLANE = ROWS * kRowWords;
  static constexpr int KP = layer_knot_pad(C1 + 3, GEN);
  static constexpr int nk = C1 + 3;
  static constexpr int kStepVecs = 4 * 2 * 64;             // u32x4 per k-step of W planes
  static constexpr int kStageVecs = KS * kStepVecs;

  // One stage = KS k-steps of 4 regions; region (ks, n) multiplies the 4 row tiles of the wave with ONE tile of B
  // operands (outputs 16n .. 16n+15, both planes): 12 matrix instructions on 4 different accumulators.  A B tile is
  // needed in one region only, so the tiles come through a ring of 4, each fetched three regions before its use
  // (with all 4 tiles of a k-step in every region the last of them could be fetched one region ahead at most, and
  // the waves waited for it).  The A operands of a k-step (4 row tiles) are read during the k-step before, one per
  // region, into the other half of A.  On entry: the E tile holds this stage, A[FLIP] the row tiles of its first
  // k-step, the ring the B tiles of its first three regions, xraw the input jets of the NEXT stage's pairs.
  //   region 0            basis of the next stage's pairs (scaled)
  //   region 1            fetch the input jets of the stage after the next; cut columns 0, 1
  //   region 2, 3         cut columns 2, 3; 4
  //   region 4 (KS - 1)   every A operand of this stage has been requested: park the next stage's planes; from here
  //                       on the reads go to the next stage's first k-step
  // wsrc: W planes of this stage (region i at wsrc + i * 128); wnext: of the next stage.
  template <bool FLIP>
  static __device__ __forceinline__ void stage(f32x4 (&acc)[4][4], WTile16 (&Wr)[4], ATile16 (&A)[2][4], f32x4& xraw,
                                               unsigned* sE, const float* knots_next, const JetScales& sc, int& prev_c0,
                                               const u32x4* __restrict__ wsrc, const u32x4* __restrict__ wnext,
                                               const char* __restrict__ xnext2, const int (&aoff)[KS][2], int lane) {
    const int p_local = lane & 15, f_local = lane >> 4;
    constexpr int kPark = 4 * (KS - 1);
    PairCols pj;
    Planes16 pp[5];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int ks = i >> 2, n = i & 3;
      const int cb = ((ks & 1) != 0) != FLIP ? 1 : 0;      // which half of A holds this k-step
      {                                                    // B tile three regions ahead, into the slot the last region left
        const int t = i + 3;
        const u32x4* src = (t < NR) ? wsrc + t * 128 : wnext + (t - NR) * 128;
        Wr[t & 3].h = src[lane];
        Wr[t & 3].l = src[64 + lane];
      }
      if (i == 0) {
        const FeatGrid fg = {knots_next[0], knots_next[nk - 1], fabsf(knots_next[GEN ? nk + 4 : nk])};
        expand_scaled<ACT, GEN>(knots_next, nk, fg, xraw, sc, pj);
      }
      if (i == 1) {
        xraw = *reinterpret_cast<const f32x4*>(xnext2);
        pp[0] = split_jet16(pj.col[0]);
        pp[1] = split_jet16(pj.col[1]);
      }
      if (i == 2) {
        pp[2] = split_jet16(pj.col[2]);
        pp[3] = split_jet16(pj.col[3]);
      }
      if (i == 3) pp[4] = split_jet16(pj.col[4]);
      if (i == kPark) {
        // lgkmcnt counts to 15 only: behind the writes below, a wait for any OLDER read would have to sit out most
        // of the writes too.  The reads in flight were requested a region or more ago: retire them first.
        __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0)
        unsigned* base = sE + (f_local * C1) * kRowWords + p_local * 2;
        park_pair16<C1, PLANE>(base, sE + (f_local * C1) * kRowWords + 32, pp, pj.c0, prev_c0);
      }
      {                                                    // row tile n of the next k-step (of the next stage after the park)
        const int kn = (ks + 1 < KS) ? ks + 1 : 0;
        A[cb ^ 1][n] = read_atile16<PLANE>(sE, aoff[kn][0], aoff[kn][1], n);
      }
      const f16x8 Bh = __builtin_bit_cast(f16x8, Wr[i & 3].h), Bl = __builtin_bit_cast(f16x8, Wr[i & 3].l);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[cb][m].l), Bh, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[cb][m].h), Bl, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[cb][m].h), Bh, acc[m][n], 0, 0, 0);
      if (i == 0) interleave<12, 8>(); else interleave<12, 4>();
      PDEKAN_REGION_FENCE();
    }
  }
};

// LONGK: layers with many input features.  The matrix instructions' accumulate is not round-to-nearest: a sum kept
// in their accumulators drifts with the number of instructions chained on it (measured on a 256-input grid-20
// layer, 576 in a row: 1.5e-5 of the result).  LONGK moves the running sums into a second set of registers with
// vector adds every kFwdFlushStages stages (one wave per SIMD: 64 more registers); narrower layers chain at most
// 128 instructions and keep two waves per SIMD.
constexpr int kFwdFlushStages = 8;
constexpr int kFwdChainMax = 128;            // matrix instructions chained on one accumulator without LONGK

// 0: some knot row of the layer is far from equal spacing (the kernel launched beside does the layer); 1: every row
// equally spaced; 2: at least nearly so.  Block-uniform.
__device__ __forceinline__ int layer_knot_class(const float* __restrict__ ginfo, int I, int lane) {
  bool near = true, equal = true;
  for (int i = lane; i < ((I + 63) & ~63); i += 64) {
    const float gi = ginfo[i < I ? i : 0];
    near = near && (__builtin_amdgcn_ballot_w64(gi != 0.0f) == ~0ull);
    equal = equal && (__builtin_amdgcn_ballot_w64(gi > 0.0f) == ~0ull);
  }
  return !near ? 0 : (equal ? 1 : 2);
}

template <int C1, int ACT, bool GEN, bool LONGK>
__device__ __forceinline__ void
fwd_f16_body(const f32x4* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
             const u32x4* __restrict__ wpl, f32x4* __restrict__ y, int64_t N, int I, int O,
             const unsigned* __restrict__ amax_in, const unsigned* __restrict__ wmax, unsigned* __restrict__ amax_out) {
  using F = FwdF16<C1, ACT, GEN>;
  constexpr int KS = F::KS, KP = F::KP, nk = F::nk, CP = F::CP;
  constexpr int kWaveWords = 2 * F::PLANE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  float* sK = reinterpret_cast<float*>(smem);                               // [I][KP], shared by the block
  unsigned* sE = reinterpret_cast<unsigned*>(smem) + I * KP + wave * kWaveWords;   // this wave's E tile
  if constexpr (GEN) {                       // [g_0 x3][the nk knots][g_last x4][1/h]
    for (int idx = tid; idx < I * KP; idx += 256) {
      const int f = idx / KP, k = idx - f * KP - 3;
      sK[idx] = (k == nk + 4) ? ginfo[f] : grid[(size_t)f * nk + (k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k))];
    }
  } else {
    for (int idx = tid; idx < I * (nk + 1); idx += 256) {
      const int f = idx / (nk + 1), k = idx - f * (nk + 1);
      sK[f * KP + k] = (k < nk) ? grid[(size_t)f * nk + k] : ginfo[f];
    }
  }
  for (int idx = lane; idx < kWaveWords; idx += 64) sE[idx] = 0u;
  const JetScales sc = expanded_jet_scales(amax_in, max_inv_h(ginfo, I, lane));
  const float sw = pow2_scale(__uint_as_float(wmax[0]));
  __syncthreads();

  const int r = lane & 15, kb = lane >> 4, p_local = lane & 15, f_local = lane >> 4;
  const int64_t P0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  const int o0 = blockIdx.y * 64;
  const int nst = I / 4;
  // E tile rows of this lane's transposed reads: lane 4q+p of a 16-lane group addresses row (column) q, (p, j) 4p .. 4p+3
  int aoff[KS][2];
  {
    const int q_ = (lane & 15) >> 2, p_ = lane & 3;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int col = ks * 32 + kstep_col(kb, half * 4 + q_);
        const int row = ((col % CP) < C1) ? (col / CP) * C1 + (col % CP) : F::ROWS - 1;
        aoff[ks][half] = row * kRowWords + 2 * p_;
      }
  }
  int64_t prow = P0 + p_local;                                              // this lane's point when it expands
  if (prow > N - 1) prow = N - 1;
  const char* xrow = reinterpret_cast<const char*>(a + prow * I + f_local);
  const u32x4* wbase = wpl + (size_t)blockIdx.y * nst * F::kStageVecs;
  const float* kbase = sK + f_local * KP + (GEN ? 3 : 0);

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (P0 < N) {                              // wave-uniform; no barrier below
    WTile16 Wr[4];
    ATile16 A[2][4];
    int prev_c0 = -100;
    f32x4 xraw;
    {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(xrow);
      xraw = *reinterpret_cast<const f32x4*>(xrow + 64);                    // 4 features x 16 bytes per stage
      const FeatGrid fg = {kbase[0], kbase[nk - 1], fabsf(kbase[GEN ? nk + 4 : nk])};
      PairCols pj;
      Planes16 pp[5];
      expand_scaled<ACT, GEN>(kbase, nk, fg, x0, sc, pj);
#pragma unroll
      for (int m = 0; m < 5; ++m) pp[m] = split_jet16(pj.col[m]);
      unsigned* base = sE + (f_local * C1) * kRowWords + p_local * 2;
      park_pair16<C1, F::PLANE>(base, sE + (f_local * C1) * kRowWords + 32, pp, pj.c0, prev_c0);
#pragma unroll
      for (int t = 0; t < 3; ++t) { Wr[t].h = wbase[t * 128 + lane]; Wr[t].l = wbase[t * 128 + 64 + lane]; }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) A[0][m] = read_atile16<F::PLANE>(sE, aoff[0][0], aoff[0][1], m);
    f32x4 tot[LONGK ? 4 : 1][LONGK ? 4 : 1];
    if constexpr (LONGK) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) tot[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int s = 0; s < nst; s += 2) {
      // stages past the end: any valid data, expanded and parked but never multiplied
      const int s1 = s + 1, s2 = (s + 2 < nst) ? s + 2 : 0, s3 = (s + 3 < nst) ? s + 3 : 0;
      F::template stage<false>(acc, Wr, A, xraw, sE, kbase + 4 * s1 * KP, sc, prev_c0,
                               wbase + (size_t)s * F::kStageVecs, wbase + (size_t)s1 * F::kStageVecs,
                               xrow + (size_t)s2 * 64, aoff, lane);
      F::template stage<(KS & 1) != 0>(acc, Wr, A, xraw, sE, kbase + 4 * s2 * KP, sc, prev_c0,
                                       wbase + (size_t)s1 * F::kStageVecs, wbase + (size_t)s2 * F::kStageVecs,
                                       xrow + (size_t)s3 * 64, aoff, lane);
      if constexpr (LONGK) {
        if (((s + 2) % kFwdFlushStages) == 0) {
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) { tot[m][n] += acc[m][n]; acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
      }
    }
    if constexpr (LONGK) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] += tot[m][n];
    }
    // D[(p, j)][output]: this lane holds the 4 jets of point 4m + kb for output 16n + r
    f32x4 inv;
#pragma unroll
    for (int j = 0; j < 4; ++j) inv[j] = rcp_pow2(sc.s[j] * sw);
    f32x4 vmax = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int64_t p = P0 + 4 * m + kb;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const f32x4 v = acc[m][n] * inv;
        if (p < N) {
          y[p * O + o0 + 16 * n + r] = v;
#pragma unroll
          for (int j = 0; j < 4; ++j) vmax[j] = fmaxf(vmax[j], fabsf(v[j]));
        }
      }
    }
    if (amax_out) publish_jet_max(vmax, amax_out, lane);
  }
}

// One launch for both variants (a second launch that only finds out it has nothing to do still costs 12-20 us at
// 2^20 points: its blocks ask for the same 146 KB of LDS and pass through the CUs one at a time): the block takes the
// variant that matches the layer's knot rows, or returns when a row is far from equal spacing.
template <int C1, int ACT, bool LONGK>
__global__ void __launch_bounds__(256, LONGK ? 1 : 2)
fwd_f16_kernel(const f32x4* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
               const u32x4* __restrict__ wpl, f32x4* __restrict__ y, int64_t N, int I, int O,
               const unsigned* __restrict__ amax_in, const unsigned* __restrict__ wmax, unsigned* __restrict__ amax_out) {
  const int cls = layer_knot_class(ginfo, I, threadIdx.x & 63);
  if (cls == 1) fwd_f16_body<C1, ACT, false, LONGK>(a, grid, ginfo, wpl, y, N, I, O, amax_in, wmax, amax_out);
  else if (cls == 2) fwd_f16_body<C1, ACT, true, LONGK>(a, grid, ginfo, wpl, y, N, I, O, amax_in, wmax, amax_out);
}

// ==========================================================================================================
// forward on the structured-sparsity matrix instruction.  Of the 16 padded columns of a feature a row of E holds 5
// (the 4 live bases c0 .. c0+3 and the base-activation column 13).  Laid out as column c -> (group c % 4, slot c / 4)
// the 4 live columns fall into 4 different groups of 4 and the base column into group 1: at most 2 non-zeros per
// group of 4, which is what v_smfmac_f32_16x16x64_f16 takes as its A operand - 8 values per lane and row for 16
// dense positions, each with a 2-bit position inside its group - at the price of a dense 16 x 16 x 32 instruction.
// Half the matrix instructions, and NO E tile: a row tile is one jet of the wave's 16 points, lane (p, f) of the
// expansion holds exactly the A operand of row p for feature f of the stage (K = 64 = 4 features x 16 positions), so
// the operands go from the expansion's registers into the instruction (a byte permute puts jet j of two columns into
// one word).  Operand layout of the instruction as measured on the hardware: A lane (row = lane & 15, ta = lane >> 4),
// compressed element j -> pair g = j / 2 of 4 dense positions, which B holds in lane block 2 (ta & 1) + g / 2,
// elements 8 (ta >> 1) + 4 (g & 1) + position; positions from bits 2j+1 : 2j of the index register.
// B operands (pack_wplanes16_kernel<14, false, true>) stream from L2 one stage ahead; D: lane (output 16n + lane & 15)
// holds points 4 (lane >> 4) .. + 3 of row tile (jet) j.
// ==========================================================================================================
typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));

struct SpA { u32x4 h[4], l[4]; unsigned idx; };            // A operands of one stage: [jet], and the positions
struct SpW { u32x4 h[2], l[2]; };                         // B operands of one (stage, output tile): [half]

__device__ __forceinline__ f16x16 sp_b16(const u32x4& a, const u32x4& b) {
  return __builtin_shufflevector(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10,
                                 11, 12, 13, 14, 15);
}

// x in group order: out[g] = in[(g - r) & 3]
__device__ __forceinline__ void sp_rotate(float (&x)[4], int r) {
  const bool r1 = (r & 1) != 0, r2 = (r & 2) != 0;
  const float a0 = r1 ? x[3] : x[0], a1 = r1 ? x[0] : x[1], a2 = r1 ? x[1] : x[2], a3 = r1 ? x[2] : x[3];
  x[0] = r2 ? a2 : a0; x[1] = r2 ? a3 : a1; x[2] = r2 ? a0 : a2; x[3] = r2 ? a1 : a3;
}

template <int C1, int ACT, bool GEN>
__device__ __forceinline__ void sp_expand(const float* knots, int nk, f32x4 jet, const JetScales& sc, SpA& out) {
  constexpr int C = C1 - 1;
  const FeatGrid fg = {knots[0], knots[nk - 1], fabsf(knots[GEN ? nk + 4 : nk])};
  Basis bs;
  if constexpr (GEN) eval_basis_near_uniform(knots, nk, fg.g0, fg.glast, fg.inv_h, jet[0], bs);
  else eval_basis_uniform_by_quotient(knots, nk, fg.g0, fg.glast, fg.inv_h, jet[0], bs);
  const Act av = eval_act(ACT, jet[0]);
  // group g holds the live column c_g = 4 slot_g + g, slot_g = (c0 >> 2) + (g < (c0 & 3))
  const int r = bs.c0 & 3, base4 = bs.c0 >> 2;
  sp_rotate(bs.b, r); sp_rotate(bs.d1, r); sp_rotate(bs.d2, r);
  // A live column outside 0 .. C-1 is column -3 .. -1 or 13 .. 15: never one of group 0; in groups 2 and 3 its
  // position, slot & 3 = 3, is that of the padding columns 14 and 15, whose weights are zero; in group 1 it is the base
  // column's position: only there the values are zeroed.
  unsigned idx = 0xCCCCu;                                  // the second element of every pair: position 3
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int slot = base4 + (g < r ? 1 : 0);
    if (g == 1) {
      const bool ok = slot >= 0 && 4 * slot + g < C;
      bs.b[g] = ok ? bs.b[g] : 0.0f; bs.d1[g] = ok ? bs.d1[g] : 0.0f; bs.d2[g] = ok ? bs.d2[g] : 0.0f;
    }
    idx |= (unsigned)(slot & 3) << (4 * g);
  }
  out.idx = idx;
  const float a1 = jet[1] * sc.s[1], a2 = jet[2] * sc.s[2], a3 = jet[3] * sc.s[3], q3 = jet[1] * jet[1] * sc.s[3];
  // word g of jet j: the live column's jet j in the low half, the pair's second element in the high half (group 1: the
  // base column, C = 4 * 3 + 1; the other groups: zero) - cut in that arrangement, so nothing is moved afterwards
  static_assert(C == 13, "the base column sits in group 1, slot 3");
  const f32x4 cb = {av.f0 * sc.s[0], av.f1 * a1, av.f1 * a2, av.f2 * q3 + av.f1 * a3};
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 cg = {bs.b[g] * sc.s[0], bs.d1[g] * a1, bs.d1[g] * a2, bs.d2[g] * q3 + bs.d1[g] * a3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned h, l;
      if (g == 1) {
        split_pair16(cg[j], cb[j], h, l);
      } else {                                 // (x, 0): the high half of the low plane's word is zero since the kernel's
        asm("v_cvt_pk_f16_f32 %0, %1, 0" : "=v"(h) : "v"(cg[j]));                       // start and is never written
        l = out.l[j][g];
        asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(cg[j]));
      }
      out.h[j][g] = h;
      out.l[j][g] = l;
    }
  }
}

constexpr int kSpStageVecs = 4 * 2 * 2 * 64;               // u32x4 of W planes per stage and block of 64 outputs

// output tile n of a stage's W planes
__device__ __forceinline__ void sp_load_w(SpW& W, const u32x4* __restrict__ stage, int n, int lane) {
  const u32x4* src = stage + n * 4 * 64 + lane;
  W.h[0] = src[0]; W.h[1] = src[64]; W.l[0] = src[128]; W.l[1] = src[192];
}

// the 4 row tiles (jets) against output tile n: 12 matrix instructions on 4 accumulators
__device__ __forceinline__ void sp_multiply(f32x4 (&acc)[4][4], const SpA& A, const SpW& W, int n) {
  const f16x16 Bh = sp_b16(W.h[0], W.h[1]), Bl = sp_b16(W.l[0], W.l[1]);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    acc[j][n] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(__builtin_bit_cast(f16x8, A.l[j]), Bh, acc[j][n], (int)A.idx, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    acc[j][n] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(__builtin_bit_cast(f16x8, A.h[j]), Bl, acc[j][n], (int)A.idx, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    acc[j][n] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(__builtin_bit_cast(f16x8, A.h[j]), Bh, acc[j][n], (int)A.idx, 0, 0);
}

// One stage: the 4 output tiles are multiplied with the B tiles in Wr; as soon as the matrix instructions of a tile
// are issued its registers take the same tile of the NEXT stage, so that all four fetches are on their way before
// the pairs of the next stage are expanded into the A operand registers (the other wave of the SIMD multiplies
// meanwhile): every fetch has the whole expansion to arrive.
template <int C1, int ACT, bool GEN>
__device__ __forceinline__ void sp_stage(f32x4 (&acc)[4][4], SpW (&Wr)[4], SpA& A, f32x4& xn,
                                         const u32x4* __restrict__ wnext, const float* knots_next, int nk,
                                         const f32x4* __restrict__ xnext2, const JetScales& sc, int lane) {
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    sp_multiply(acc, A, Wr[n], n);
    sp_load_w(Wr[n], wnext, n, lane);
    PDEKAN_REGION_FENCE();
  }
  sp_expand<C1, ACT, GEN>(knots_next, nk, xn, sc, A);
  xn = *xnext2;
  PDEKAN_REGION_FENCE();
}

template <int C1, int ACT, bool GEN, bool LONGK>
__device__ __forceinline__ void
fwd_sp_body(const f32x4* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
            const u32x4* __restrict__ wsp, f32x4* __restrict__ y, int64_t N, int I, int O,
            const unsigned* __restrict__ amax_in, const unsigned* __restrict__ wmax, unsigned* __restrict__ amax_out) {
  constexpr int nk = C1 + 3, KP = layer_knot_pad(nk, GEN);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  float* sK = reinterpret_cast<float*>(smem);                               // [I][KP], shared by the block
  if constexpr (GEN) {                       // [g_0 x3][the nk knots][g_last x4][1/h]
    for (int idx = tid; idx < I * KP; idx += 256) {
      const int f = idx / KP, k = idx - f * KP - 3;
      sK[idx] = (k == nk + 4) ? ginfo[f] : grid[(size_t)f * nk + (k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k))];
    }
  } else {
    for (int idx = tid; idx < I * (nk + 1); idx += 256) {
      const int f = idx / (nk + 1), k = idx - f * (nk + 1);
      sK[f * KP + k] = (k < nk) ? grid[(size_t)f * nk + k] : ginfo[f];
    }
  }
  const JetScales sc = expanded_jet_scales(amax_in, max_inv_h(ginfo, I, lane));
  const float sw = pow2_scale(__uint_as_float(wmax[0]));
  __syncthreads();

  const int oc = lane & 15, tb = lane >> 4;                // expansion: point oc of the wave, feature tb of the stage
  const int o0 = blockIdx.y * 64;
  const int nst = I / 4;
  const u32x4* wbase = wsp + (size_t)blockIdx.y * nst * kSpStageVecs;
  const float* kbase = sK + tb * KP + (GEN ? 3 : 0);
  // A block walks tiles of 64 points (16 per wave), blockIdx.x, + gridDim.x, ...: the stages of a wave's tiles
  // form one stream - the last stages of a tile expand the first pairs of the next tile and fetch its first B tiles
  // (the same for every tile) and input jets, so that only the first tile of a wave starts cold.
  auto row_of = [&](int64_t tile) {
    int64_t prow = (tile * 4 + wave) * 16 + oc;
    if (prow > N - 1) prow = N - 1;
    return a + prow * I + tb;                              // + 4 per stage
  };
  int64_t tile = blockIdx.x;
  if ((tile * 4 + wave) * 16 >= N) return;                 // wave-uniform; no barrier below
  const f32x4* xrow = row_of(tile);

  f32x4 acc[4][4], tot[LONGK ? 4 : 1][LONGK ? 4 : 1];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (LONGK) tot[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  f32x4 inv;
#pragma unroll
  for (int j = 0; j < 4; ++j) inv[j] = rcp_pow2(sc.s[j] * sw);
  f32x4 vmax = {0.f, 0.f, 0.f, 0.f};
  // stage s is multiplied, then the pairs of stage s + 1 are expanded; the input jets of stage s + 2 are on their way
  SpW Wr[4];
  SpA A;
#pragma unroll
  for (int j = 0; j < 4; ++j) A.l[j] = (u32x4){0u, 0u, 0u, 0u};     // (sp_expand keeps the high halves)
  f32x4 xn;
#pragma unroll
  for (int n = 0; n < 4; ++n) sp_load_w(Wr[n], wbase, n, lane);
  sp_expand<C1, ACT, GEN>(kbase, nk, xrow[0], sc, A);
  xn = xrow[4];                                            // (at least 2 stages: I is a multiple of 8)
  for (;;) {
    const int64_t P0 = (tile * 4 + wave) * 16;
    const int64_t tnext = tile + gridDim.x;
    const bool more = (tnext * 4 + wave) * 16 < N;
    const f32x4* xrow_next = more ? row_of(tnext) : xrow;  // (past the last tile: any valid data, never multiplied)
    for (int s = 0; s < nst; s += 2) {                     // (an even number of stages)
      const int s1 = s + 1, s2 = (s + 2 < nst) ? s + 2 : s + 2 - nst, s3 = (s + 3 < nst) ? s + 3 : s + 3 - nst;
      const f32x4* x2 = ((s + 2 < nst) ? xrow : xrow_next) + 4 * s2;
      const f32x4* x3 = ((s + 3 < nst) ? xrow : xrow_next) + 4 * s3;
      sp_stage<C1, ACT, GEN>(acc, Wr, A, xn, wbase + (size_t)s1 * kSpStageVecs, kbase + 4 * s1 * KP, nk, x2, sc, lane);
      sp_stage<C1, ACT, GEN>(acc, Wr, A, xn, wbase + (size_t)s2 * kSpStageVecs, kbase + 4 * s2 * KP, nk, x3, sc, lane);
      if constexpr (LONGK) {
        if (((s + 2) % kFwdFlushStages) == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 4; ++n) { tot[j][n] += acc[j][n]; acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
      }
    }
    // D: row tile j = jet j of the wave's 16 points; this lane holds points 4 tb + i for output 16n + oc
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t p = P0 + 4 * tb + i;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        f32x4 v = {acc[0][n][i], acc[1][n][i], acc[2][n][i], acc[3][n][i]};
        if constexpr (LONGK) v += (f32x4){tot[0][n][i], tot[1][n][i], tot[2][n][i], tot[3][n][i]};
        v *= inv;
        if (p < N) {
          y[p * O + o0 + 16 * n + oc] = v;
#pragma unroll
          for (int j = 0; j < 4; ++j) vmax[j] = fmaxf(vmax[j], fabsf(v[j]));
        }
      }
    }
    if (!more) break;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (LONGK) tot[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    tile = tnext;
    xrow = xrow_next;
  }
  if (amax_out) publish_jet_max(vmax, amax_out, lane);
}

template <int C1, int ACT, bool LONGK>
__global__ void __launch_bounds__(256, LONGK ? 1 : 2)
fwd_sp_f16_kernel(const f32x4* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
                  const u32x4* __restrict__ wsp, f32x4* __restrict__ y, int64_t N, int I, int O,
                  const unsigned* __restrict__ amax_in, const unsigned* __restrict__ wmax, unsigned* __restrict__ amax_out) {
  const int cls = layer_knot_class(ginfo, I, threadIdx.x & 63);
  if (cls == 1) fwd_sp_body<C1, ACT, false, LONGK>(a, grid, ginfo, wsp, y, N, I, O, amax_in, wmax, amax_out);
  else if (cls == 2) fwd_sp_body<C1, ACT, true, LONGK>(a, grid, ginfo, wsp, y, N, I, O, amax_in, wmax, amax_out);
}

// ==========================================================================================================
// input-jet gradients.  gE[(p, j)][k] = sum_o gY[(p, j)][o] * W[o][k]: M = (point, jet), K = 64 outputs (2 k-steps),
// N = 16 columns.  Wave = 16 points; the A operands (the wave's gY rows, cut into planes once) stay in registers,
// the B operands (W planes) stream from L2 one column tile at a time, three regions ahead.  A stage = 4 features =
// TA or TB column tiles (BwiTiles: the 8 * C1 columns of a pair of stages, unpadded) x 2 k-steps = regions of 12
// matrix instructions; its tile of gE is parked in LDS at column slot g % RING (an even stage's last tile may hold
// the first columns of the odd stage's first feature: the odd stage leaves those slots alone), every
// (point, feature) lane pulls the 4+1 live columns of its pair and applies the chain rule through
// B', B'', B''' and the activation derivatives - during the matrix instructions of the NEXT stage.
// Layers wider than 64 outputs are walked 64 outputs at a time; each pass adds its share to the gradients the
// same lane stored at the same address in the pass before (the chain rule is linear in gE).
// ==========================================================================================================
template <int C1, int ACT, bool GEN>
struct BwiF16 {
  static constexpr int ROW = BwiTiles<C1>::RING * 4 + 4;   // words per point in the parked gE tile (+4: bank spread)
  static constexpr int GROW = 64 * 4 + 4;                  // words per point of the staged raw gY rows
  static constexpr int TILE = 16 * (ROW > GROW ? ROW : GROW);
  static constexpr int KP = layer_knot_pad(C1 + 3, GEN);
  static constexpr int nk = C1 + 3;
};

// basis with the third derivative for the chain rule: the interval is settled by comparisons against the knots,
// with one round of knot reads
template <int ACT, bool GEN>
__device__ __forceinline__ void eval_pair_d3(const float* knots, int nk, const FeatGrid& fg, float x, Basis& bs, Act& av) {
  if constexpr (GEN) eval_basis_near_uniform<true>(knots, nk, fg.g0, fg.glast, fg.inv_h, x, bs);
  else eval_basis_uniform_oneshot<true>(knots, nk, fg.g0, fg.glast, fg.inv_h, x, bs);
  av = eval_act(ACT, x);
}

// what the chain rule of one (point, feature) pair needs once its gE columns are there
struct PairBack { Basis bs; Act av; f32x4 x; };

// gradient with respect to the input jet of one pair from the gE jets q of its 4 + 1 live columns.  The sums over the
// columns come first (whole jets at a time: packed FMAs), then the scales are taken out and the jets combined:
//   d/da_0 = U1_0 + a_1 U2_1 + a_2 U2_2 + a_3 U2_3 + a_1^2 U3_3     d/da_1 = U1_1 + 2 a_1 U2_3
//   d/da_2 = U1_2                                                     d/da_3 = U1_3
// with U1, U2, U3 = sum over the columns of q times the first, second, third derivative of the column's basis
// (of the activation for the base column).
__device__ __forceinline__ f32x4 chain_rule(const PairBack& pb, const f32x4 (&dcol)[5], const f32x4& inv) {
  f32x4 U1 = {0.f, 0.f, 0.f, 0.f}, U2 = {0.f, 0.f, 0.f, 0.f};
  float U3 = 0.f;
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const float e1 = m < 4 ? pb.bs.d1[m & 3] : pb.av.f1;
    const float e2 = m < 4 ? pb.bs.d2[m & 3] : pb.av.f2;
    const float e3 = m < 4 ? pb.bs.d3[m & 3] : pb.av.f3;
    U1 += dcol[m] * e1;
    U2 += dcol[m] * e2;
    U3 += dcol[m][3] * e3;
  }
  U1 *= inv;
  U2 *= inv;
  U3 *= inv[3];
  f32x4 out = U1;
  out[0] += pb.x[1] * U2[1] + pb.x[2] * U2[2] + pb.x[3] * U2[3] + pb.x[1] * pb.x[1] * U3;
  out[1] += 2.0f * pb.x[1] * U2[3];
  return out;
}

// N must be a multiple of 16 (whole waves; the caller gives the last N % 16 points to the sparse kernel).
// MULTI: more than 64 outputs (several passes, read-modify-write of ga).
template <int C1, int ACT, bool GEN, bool MULTI>
__global__ void __launch_bounds__(256, 2)
bwd_input_f16_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ gy, const float* __restrict__ grid,
                     const float* __restrict__ ginfo, const u32x4* __restrict__ wpl, f32x4* __restrict__ ga, int64_t N,
                     int I, int O, const unsigned* __restrict__ cot_in, const unsigned* __restrict__ wmax,
                     unsigned* __restrict__ cot_out) {
  using F = BwiF16<C1, ACT, GEN>;
  using T = BwiTiles<C1>;
  constexpr int KP = F::KP, nk = F::nk, ROW = F::ROW, GROW = F::GROW;
  constexpr int C = C1 - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // (the two variants stay separate launches here: with both bodies in one kernel, as in the forward, the grid-20
  // instantiations keep a pair's basis in scratch memory and lose 5 %)
  if (layer_knot_class(ginfo, I, lane) != (GEN ? 2 : 1)) return;
  float* sK = reinterpret_cast<float*>(smem);                               // [I][KP], shared by the block
  float* sD = sK + I * KP + wave * F::TILE;                                 // this wave's tile
  if constexpr (GEN) {
    for (int idx = tid; idx < I * KP; idx += 256) {
      const int f = idx / KP, k = idx - f * KP - 3;
      sK[idx] = (k == nk + 4) ? ginfo[f] : grid[(size_t)f * nk + (k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k))];
    }
  } else {
    for (int idx = tid; idx < I * (nk + 1); idx += 256) {
      const int f = idx / (nk + 1), k = idx - f * (nk + 1);
      sK[f * KP + k] = (k < nk) ? grid[(size_t)f * nk + k] : ginfo[f];
    }
  }
  // per-jet scales of the cotangent rows, the weights' scale, and what divides both out of gE
  float tj[4];
  f32x4 inv;
  {
    const float sw = pow2_scale(__uint_as_float(wmax[0]));
#pragma unroll
    for (int j = 0; j < 4; ++j) { tj[j] = pow2_scale(__uint_as_float(cot_in[j])); inv[j] = rcp_pow2(tj[j] * sw); }
  }
  __syncthreads();

  const int r = lane & 15, kb = lane >> 4, p_local = lane & 15, f_local = lane >> 4;
  const int64_t P0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  if (P0 >= N) return;                        // wave-uniform; no barrier below
  const int nst = I / 4;
  const int64_t prow = P0 + p_local;
  const char* xrow = reinterpret_cast<const char*>(a + prow * I + f_local);
  f32x4* orow = ga + prow * I + f_local;
  const float* kbase = sK + f_local * KP + (GEN ? 3 : 0);
  const float my_t = tj[r & 3];               // this lane's A-operand row is jet r & 3 of point r >> 2 of the m-tile
  f32x4 vmax = {0.f, 0.f, 0.f, 0.f};
  const int npass = MULTI ? O / 64 : 1;

  for (int ob = 0; ob < npass; ++ob) {
    // ---- the wave's gY rows of this output block -> A operands in registers ----
    u32x4 GAh[4][2], GAl[4][2];
    {
      f32x4 graw[16];                         // 16 points x 64 outputs x 16 bytes, 1 KiB per instruction, all in flight
#pragma unroll
      for (int v = 0; v < 16; ++v) graw[v] = gy[(P0 + v) * O + ob * 64 + lane];
#pragma unroll
      for (int v = 0; v < 16; ++v) *reinterpret_cast<f32x4*>(sD + v * GROW + lane * 4) = graw[v];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const float* src = sD + (4 * m + (r >> 2)) * GROW + (32 * ks + 8 * kb) * 4 + (r & 3);
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = src[e * 4] * my_t;
          unsigned h[4], l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split_pair16(v[2 * e], v[2 * e + 1], h[e], l[e]);
          GAh[m][ks] = (u32x4){h[0], h[1], h[2], h[3]};
          GAl[m][ks] = (u32x4){l[0], l[1], l[2], l[3]};
        }
    }
    const u32x4* wbase = wpl + (size_t)ob * (nst / 2) * T::kPairVecs;
    // region i of a stage of NBs column tiles: k-step i / NBs, column tile i % NBs; its B operand sits at
    // wsrc + i * 128 + lane
    WTile16 Wt[4];
    auto fetch_w = [&](WTile16& w, const u32x4* src) { w.h = src[lane]; w.l = src[64 + lane]; };
    fetch_w(Wt[0], wbase);
    fetch_w(Wt[1], wbase + 128);
    fetch_w(Wt[2], wbase + 256);

    f32x4 acc[4][T::TA];
    // pipeline registers: `cur` = the pair of the stage being multiplied, `prev` = the pair of the stage before with
    // its gE columns.  Before the first stage `prev` is all zeros: its chain rule then rewrites what is already there.
    PairBack prev, cur;
    f32x4 dcol[5];
    f32x4 old = {0.f, 0.f, 0.f, 0.f};
    if (MULTI && ob > 0) old = orow[0];
    prev.x = (f32x4){0.f, 0.f, 0.f, 0.f};
    prev.av = Act{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) prev.bs.d1[m] = prev.bs.d2[m] = prev.bs.d3[m] = 0.0f;
#pragma unroll
    for (int m = 0; m < 5; ++m) dcol[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    cur.x = *reinterpret_cast<const f32x4*>(xrow);
    f32x4 xnext = (nst > 1) ? *reinterpret_cast<const f32x4*>(xrow + 64) : cur.x;

    // one stage: NBs column tiles, B-tile ring starting at slot PH, ODD: the second stage of its pair
    auto stage = [&](auto nb_c, auto ph_c, auto odd_c, int s, const u32x4* wsrc, const u32x4* wnxt)
                     __attribute__((always_inline)) {
      constexpr int NBs = decltype(nb_c)::value, PH = decltype(ph_c)::value, NRs = 2 * NBs;
      constexpr bool ODD = decltype(odd_c)::value;
      f32x4* odst = orow + (s > 0 ? s - 1 : 0) * 4;
#pragma unroll
      for (int i = 0; i < NRs; ++i) {
        const int ks = i / NBs, n = i % NBs;
        {                                     // B operand three regions ahead (the slot the last region has just left)
          const int t = i + 3;
          fetch_w(Wt[(t + PH) & 3], (t < NRs) ? wsrc + t * 128 : wnxt + (t - NRs) * 128);
        }
        if (i == 0) {                         // chain rule of the stage before, on the columns read after its park
          f32x4 out = chain_rule(prev, dcol, inv);
          if (MULTI) out += old;
          *odst = out;
          if (!MULTI || ob == npass - 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) vmax[j] = fmaxf(vmax[j], fabsf(out[j]));
          }
        }
        if (i == 1) {                         // this stage's pair: basis, activation (kept for the next stage)
          const float* kn = kbase + 4 * s * KP;
          const FeatGrid fg = {kn[0], kn[nk - 1], fabsf(kn[GEN ? nk + 4 : nk])};
          eval_pair_d3<ACT, GEN>(kn, nk, fg, cur.x[0], cur.bs, cur.av);
          if (MULTI) old = (ob > 0) ? orow[s * 4] : old;
        }
        if (i > NBs) {                        // the column tile the region before completed: park it (see below)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            *reinterpret_cast<f32x4*>(sD + (4 * m + kb) * ROW + (16 * (n - 1) + r) * 4) = acc[m][n - 1];
        }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};   // the first k-step starts the sums (no separate clearing pass)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, GAl[m][ks]),
                                                             __builtin_bit_cast(f16x8, Wt[(i + PH) & 3].h),
                                                             ks == 0 ? zero4 : acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, GAh[m][ks]),
                                                             __builtin_bit_cast(f16x8, Wt[(i + PH) & 3].l), acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, GAh[m][ks]),
                                                             __builtin_bit_cast(f16x8, Wt[(i + PH) & 3].h), acc[m][n], 0, 0, 0);
        if (i == 0) interleave<12, 10>(); else if (i == 1) interleave<12, 8>(); else interleave<12, 2>();
        PDEKAN_REGION_FENCE();
      }
      // park the stage's gE tile: lane (c = r, q = kb) holds the 4 jets of point 4m + q, column slot 16n + c.  The
      // tiles before the last one went out during the regions above (the previous stage's columns were read before them)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        *reinterpret_cast<f32x4*>(sD + (4 * m + kb) * ROW + (16 * (NBs - 1) + r) * 4) = acc[m][NBs - 1];
      // the 4 + 1 live columns of this lane's pair; a column outside [0, C) reads the base column's slot and is
      // zeroed by a select (no branch)
      const float* drow = sD + p_local * ROW;
      const int g0 = (ODD ? T::OFFB : 0) + f_local * C1;
      auto slot_of = [&](int c) { return T::WRAP ? (g0 + c) % T::RING : f_local * C1 + c; };
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int c = cur.bs.c0 + m;
        const bool ok = (c >= 0) & (c < C);
        const f32x4 v = *reinterpret_cast<const f32x4*>(drow + slot_of(ok ? c : C) * 4);
        dcol[m] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      dcol[4] = *reinterpret_cast<const f32x4*>(drow + slot_of(C) * 4);
      prev = cur;
      cur.x = xnext;
      xnext = *reinterpret_cast<const f32x4*>(xrow + (size_t)((s + 2 < nst) ? s + 2 : 0) * 64);
    };
    using std::integral_constant;
    if constexpr (T::TA == T::TB && T::PH1 == 0 && T::PH2 == 0 && T::OFFB % T::RING == 0) {
      // every stage is whole tiles of its own and leaves the ring where it found it: one stage body
      for (int s = 0; s < nst; ++s) {
        const u32x4* pA = wbase + (size_t)s * (T::kPairVecs / 2);
        stage(integral_constant<int, T::TA>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}, s, pA,
              (s + 1 < nst) ? pA + T::kPairVecs / 2 : wbase);
      }
    } else
    for (int s = 0; s < nst; s += 4) {          // two pairs of stages (nst is even)
      const bool more = s + 2 < nst;
      const u32x4* pA = wbase + (size_t)(s >> 1) * T::kPairVecs;
      const u32x4* pB = pA + 2 * T::TA * 128;
      const u32x4* pC = more ? pA + T::kPairVecs : wbase;            // past the end: any valid planes, never multiplied
      stage(integral_constant<int, T::TA>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}, s, pA, pB);
      stage(integral_constant<int, T::TB>{}, integral_constant<int, T::PH1>{}, integral_constant<bool, true>{}, s + 1, pB, pC);
      if (more) {
        const u32x4* pD = pC + 2 * T::TA * 128;
        const u32x4* pE = (s + 4 < nst) ? pC + T::kPairVecs : wbase;
        stage(integral_constant<int, T::TA>{}, integral_constant<int, T::PH2>{}, integral_constant<bool, false>{}, s + 2, pC, pD);
        stage(integral_constant<int, T::TB>{}, integral_constant<int, T::PH3>{}, integral_constant<bool, true>{}, s + 3, pD, pE);
      }
    }
    {                                           // chain rule of the last stage
      f32x4 out = chain_rule(prev, dcol, inv);
      if (MULTI) out += old;
      orow[(nst - 1) * 4] = out;
      if (!MULTI || ob == npass - 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) vmax[j] = fmaxf(vmax[j], fabsf(out[j]));
      }
    }
  }
  if (cot_out) publish_jet_max(vmax, cot_out, lane);
}

// ==========================================================================================================
// weight gradients.  gW[k][o] = sum over (p, j) of E[(p, j)][k] * gY[(p, j)][o]: M = 16 columns, N = 16 outputs,
// K = 32 = 8 points x 4 jets.  Every wave is on its own (no block barrier): wave = FE features x C1 columns
// (NT = FE*C1/16 column tiles: 8 x 14 at grid 10, 4 x 24 at grid 20) x 64 outputs of accumulators over a chunk of
// points, walked 8 points at a time.  One wave per SIMD (224 accumulator registers), so the instruction stream is
// software-pipelined in source: the vector work that prepares step s+1 sits between the matrix instructions of
// step s, one fenced region of 12 matrix instructions per column tile.
//   B operand (gY): lane (r = lane&15, q = lane>>4) holds output 16m + r, jets of points 2q and 2q+1: two 16-byte
//     loads per output tile, fetched two steps ahead, scaled per jet and cut into planes at the start of their step.
//   A operand (E^T): lane (p = lane&7, f = lane>>3) expands its (point, feature) pair, cuts the 20 scaled jets and
//     parks them in the wave's own LDS tile [2 planes][FE*C1 columns][32 fp16 + pad]; the 4 columns it wrote one
//     step earlier are cleared, everything else of the tile stays zero.  Read back in operand order (ds_read_b128).
// The jet index is summed over here, so the scales of the two operands are tied: gY's jet j by t_j (its own
// maximum under 2^13), E's by s_j = c / t_j with the one power of two c that keeps every s_j * max|E_j| under 2^13.
// The matrix instructions' accumulate is not round-to-nearest: they run 512 points at a time into acc, which is
// then added (vector add) into tot and cleared.
// Blocks (4 waves = 4*FE features) run in the variant that matches their features' knot rows (GEN = false: all
// equally spaced; GEN = true: all at least nearly so); bwd_weight_mfma_kernel with skip_uniform = 1, launched on the
// same grid, does the blocks both leave out.  Output: part[chunk][O][I][C1].
// ==========================================================================================================
constexpr int kF16FlushPoints = 512;
constexpr int kWgRw = 20;                    // words per E tile row: 32 fp16 + 16 bytes of pad (bank spread; dump slot)

struct GyPlanes16 { u32x4 h[4], l[4]; };     // 4 output tiles x (2 points x 4 jets)
struct TileOps16 { u32x4 h, l; };

__device__ __forceinline__ void split_gy_tile16(const f32x4 (&g)[2], const f32x4& t, GyPlanes16& G, int m) {
  const Planes16 p0 = split_jet16(g[0] * t), p1 = split_jet16(g[1] * t);
  G.h[m] = (u32x4){p0.h.x, p0.h.y, p1.h.x, p1.h.y};
  G.l[m] = (u32x4){p0.l.x, p0.l.y, p1.l.x, p1.l.y};
}

template <int C1, int COLS>
__device__ __forceinline__ void park_wg16(unsigned* base, unsigned* dump, const Planes16 (&pp)[5], int c0, int& prev_c0) {
  constexpr int C = C1 - 1, PL = COLS * kWgRw;
  const u32x2 zero = {0u, 0u};
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = prev_c0 + m;
    unsigned* dst = (c >= 0 && c < C) ? base + c * kWgRw : dump;
    *reinterpret_cast<u32x2*>(dst) = zero;
    *reinterpret_cast<u32x2*>(dst + PL) = zero;
  }
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const int c = (m < 4) ? c0 + m : C;
    unsigned* dst = (m == 4 || (c >= 0 && c < C)) ? base + c * kWgRw : dump;
    *reinterpret_cast<u32x2*>(dst) = pp[m].h;
    *reinterpret_cast<u32x2*>(dst + PL) = pp[m].l;
  }
  prev_c0 = c0;
}

template <int COLS>
__device__ __forceinline__ TileOps16 read_wg_tile(const unsigned* esrc, int t) {
  TileOps16 o;
  o.h = *reinterpret_cast<const u32x4*>(esrc + t * 16 * kWgRw);
  o.l = *reinterpret_cast<const u32x4*>(esrc + t * 16 * kWgRw + COLS * kWgRw);
  return o;
}

__device__ __forceinline__ void mfma_wg_tile(f32x4 (&acc)[4], const TileOps16& e, const GyPlanes16& G) {
  const f16x8 eh = __builtin_bit_cast(f16x8, e.h), el = __builtin_bit_cast(f16x8, e.l);
#pragma unroll
  for (int m = 0; m < 4; ++m)
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(el, __builtin_bit_cast(f16x8, G.h[m]), acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < 4; ++m)
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, __builtin_bit_cast(f16x8, G.l[m]), acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < 4; ++m)
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, __builtin_bit_cast(f16x8, G.h[m]), acc[m], 0, 0, 0);
}

__device__ __forceinline__ f32x4 ld16v(const char* __restrict__ row, unsigned off) {
  return *reinterpret_cast<const f32x4*>(row + off);
}

// One step = 8 points = NT regions of 12 matrix instructions (one column tile each).  graw / xraw hold the raw rows
// of step s+1 on entry; each piece is fetched again (rows of step s+2) right after the region that consumed it:
//   region 0          evaluate and scale the pairs of step s+1 (xraw); fetch gY tile 3 of step s+1
//   region 1          fetch xraw of step s+2; cut the pairs' columns 0, 1, 2
//   region 2          cut columns 3, 4
//   regions NT-4 ..   scale and cut gY tile 0..3 of step s+1 (Gn); regions NT-3 .. fetch tiles 0..2 of step s+2
//   region NT-3       read the operands of the last tile, then park the pairs' planes of step s+1
//   region NT-1       read the first operands of step s+1
template <int C1, int FE, int ACT, bool GEN>
__device__ __forceinline__ void wg16_step(f32x4 (&acc)[FE * C1 / 16][4], const GyPlanes16& G, GyPlanes16& Gn, TileOps16& cur,
                                          f32x4 (&graw)[4][2], f32x4& xraw, unsigned* sE, const float* knots,
                                          const FeatGrid& fg, const JetScales& se, const f32x4& tg, int& prev_c0,
                                          const char* __restrict__ xrow2, const char* __restrict__ grow1,
                                          const char* __restrict__ grow2, unsigned xoff, unsigned goff, int lane, int O) {
  constexpr int COLS = FE * C1, NT = COLS / 16, nk = C1 + 3;
  static_assert(NT == 6 || NT == 7, "the region plan below is written for 6 or 7 column tiles");
  const int r = lane & 15, q = lane >> 4, p_local = lane & 7, f_local = (lane >> 3) % FE;
  const unsigned* esrc = sE + r * kWgRw + q * 4;
  PairCols pj;
  Planes16 pp[5];
  TileOps16 last;                             // operands of the last tile, read one region early so that the park can go early
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    TileOps16 nxt;
    if (t + 1 < NT - 1) nxt = read_wg_tile<COLS>(esrc, t + 1);
    if (t == NT - 3) last = read_wg_tile<COLS>(esrc, NT - 1);
    if (t == NT - 2) nxt = last;
    if (t == 0) {
      expand_scaled<ACT, GEN>(knots, nk, fg, xraw, se, pj);
#pragma unroll
      for (int e = 0; e < 2; ++e) graw[3][e] = ld16v(grow1, goff + (unsigned)(e * O + 3 * 16) * 16u);
    }
    if (t == 1) {
      xraw = ld16v(xrow2, xoff);
      pp[0] = split_jet16(pj.col[0]); pp[1] = split_jet16(pj.col[1]); pp[2] = split_jet16(pj.col[2]);
    }
    if (t == 2) { pp[3] = split_jet16(pj.col[3]); pp[4] = split_jet16(pj.col[4]); }
    if (t >= NT - 4) split_gy_tile16(graw[t - (NT - 4)], tg, Gn, t - (NT - 4));
    if (t >= NT - 3) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
        graw[t - (NT - 3)][e] = ld16v(grow2, goff + (unsigned)(e * O + (t - (NT - 3)) * 16) * 16u);
    }
    if (t == NT - 3) {                        // every operand read of this step's E tile has been issued
      unsigned* base = sE + (f_local * C1) * kWgRw + p_local * 2;
      park_wg16<C1, COLS>(base, sE + (f_local * C1) * kWgRw + 16, pp, pj.c0, prev_c0);
    }
    if (t == NT - 1) nxt = read_wg_tile<COLS>(esrc, 0);
    mfma_wg_tile(acc[t], cur, G);
    cur = nxt;
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int C1, int FE, int ACT, bool GEN>
__global__ void __launch_bounds__(256, 1)
bwd_weight_f16_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ gy, const float* __restrict__ grid,
                      const float* __restrict__ ginfo, float* __restrict__ part, int64_t N, int I, int O,
                      int64_t chunk_pts, int nx, int ny, int nz, const unsigned* __restrict__ act_in,
                      const unsigned* __restrict__ cot_in) {
  constexpr int COLS = FE * C1, NT = COLS / 16, nk = C1 + 3;
  static_assert(COLS % 16 == 0 && (FE == 8 || FE == 4), "16-wide column tiles; 8 points x FE features per step");
  constexpr int KP = GEN ? kKnotPadN : kKnotPadU;
  constexpr int kWaveWords = 2 * COLS * kWgRw + FE * KP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bx, by, bz;                             // blocks that read the same gY rows are 8 ids apart: one XCD's L2
  {
    const int b = blockIdx.x;
    if (nz % 8 == 0) {
      const int rr = b >> 3;
      bx = rr % nx; by = (rr / nx) % ny; bz = (rr / (nx * ny)) * 8 + (b & 7);
    } else {
      bx = b % nx; by = (b / nx) % ny; bz = b / (nx * ny);
    }
  }
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  {                                           // the block's features: all equally spaced -> GEN = false takes it;
    const float gi = ginfo[bx * 4 * FE + lane % (4 * FE)];          // all at least nearly so -> GEN = true
    const bool all_near = __builtin_amdgcn_ballot_w64(gi != 0.0f) == ~0ull;
    const bool all_equal = __builtin_amdgcn_ballot_w64(gi > 0.0f) == ~0ull;
    if (!all_near || all_equal == GEN) return;
  }
  unsigned* sE = reinterpret_cast<unsigned*>(smem) + wave * kWaveWords;     // this wave's tile: [2][COLS][kWgRw]
  float* sK = reinterpret_cast<float*>(sE + 2 * COLS * kWgRw);              // and its FE knot rows
  // FE = 4: lanes 32..63 repeat the pairs of lanes 0..31 (same values to the same LDS words)
  const int r = lane & 15, q = lane >> 4, p_local = lane & 7, f_local = (lane >> 3) % FE;
  const int i0 = (bx * 4 + wave) * FE, o0 = by * 64;
  if constexpr (GEN) {                        // [g_0 x3][the nk knots][g_last x4]
    for (int idx = lane; idx < FE * KP; idx += 64) {
      const int f = idx / KP, k = idx - f * KP - 3;
      sK[idx] = grid[(size_t)(i0 + f) * nk + (k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k))];
    }
  } else {
    for (int idx = lane; idx < FE * (nk + 1); idx += 64) {
      const int f = idx / (nk + 1), k = idx - f * (nk + 1);
      sK[f * KP + k] = (k < nk) ? grid[(size_t)(i0 + f) * nk + k] : ginfo[i0 + f];
    }
  }
  for (int idx = lane; idx < 2 * COLS * kWgRw; idx += 64) sE[idx] = 0u;
  const float* knots = sK + f_local * KP + (GEN ? 3 : 0);
  const FeatGrid fg = {grid[(size_t)(i0 + f_local) * nk], grid[(size_t)(i0 + f_local) * nk + nk - 1],
                       fabsf(ginfo[i0 + f_local])};
  // scales: gY's jet j by t_j; E's jet j by c / t_j, c = the smallest of (scale that E_j alone could take) * t_j
  JetScales se;
  f32x4 tg;
  float inv_c;
  {
    const JetScales eb = expanded_jet_scales(act_in, max_inv_h(ginfo, I, lane));
    float c = 3.0e38f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tg[j] = pow2_scale(__uint_as_float(cot_in[j]));
      c = fminf(c, eb.s[j] * tg[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) se.s[j] = c * rcp_pow2(tg[j]);
    inv_c = rcp_pow2(c);
  }

  f32x4 acc[NT][4], tot[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) { acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f}; tot[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  // N and chunk_pts are multiples of 16: whole pairs of steps
  const int64_t pbeg = (int64_t)bz * chunk_pts;
  const int64_t pend = (pbeg + chunk_pts < N) ? pbeg + chunk_pts : N;
  const char* xb = reinterpret_cast<const char*>(a);
  const char* gb = reinterpret_cast<const char*>(gy);
  const unsigned xoff = (unsigned)(p_local * I + i0 + f_local) * 16u;
  const unsigned goff = (unsigned)(2 * q * O + o0 + r) * 16u;
  const size_t xstride = (size_t)I * 16, gstride = (size_t)O * 16;       // bytes per point

  if (pbeg < pend) {
    f32x4 graw[4][2], xraw;
    GyPlanes16 GA, GB;
    int prev_c0 = -100;
    {
      f32x4 g0[4][2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          g0[m][e] = ld16v(gb + pbeg * gstride, goff + (unsigned)(e * O + m * 16) * 16u);
          graw[m][e] = ld16v(gb + (pbeg + 8) * gstride, goff + (unsigned)(e * O + m * 16) * 16u);
        }
      const f32x4 x0 = ld16v(xb + pbeg * xstride, xoff);
      xraw = ld16v(xb + (pbeg + 8) * xstride, xoff);
      PairCols pj;
      Planes16 pp[5];
      expand_scaled<ACT, GEN>(knots, nk, fg, x0, se, pj);
#pragma unroll
      for (int m = 0; m < 5; ++m) pp[m] = split_jet16(pj.col[m]);
      unsigned* base = sE + (f_local * C1) * kWgRw + p_local * 2;
      park_wg16<C1, COLS>(base, sE + (f_local * C1) * kWgRw + 16, pp, pj.c0, prev_c0);
#pragma unroll
      for (int m = 0; m < 4; ++m) split_gy_tile16(g0[m], tg, GA, m);
    }
    TileOps16 cur = read_wg_tile<COLS>(sE + r * kWgRw + q * 4, 0);
    for (int64_t pb = pbeg; pb < pend;) {
      const int64_t pstop = (pb + kF16FlushPoints < pend) ? pb + kF16FlushPoints : pend;
      for (; pb < pstop; pb += 16) {
        // past the end of the chunk: any valid rows, fetched (and expanded) but never used
        const int64_t p1 = pb + 8, p2 = (pb + 16 < pend) ? pb + 16 : pbeg, p3 = (pb + 24 < pend) ? pb + 24 : pbeg;
        wg16_step<C1, FE, ACT, GEN>(acc, GA, GB, cur, graw, xraw, sE, knots, fg, se, tg, prev_c0, xb + p2 * xstride,
                                    gb + p1 * gstride, gb + p2 * gstride, xoff, goff, lane, O);
        wg16_step<C1, FE, ACT, GEN>(acc, GB, GA, cur, graw, xraw, sE, knots, fg, se, tg, prev_c0, xb + p3 * xstride,
                                    gb + p2 * gstride, gb + p3 * gstride, xoff, goff, lane, O);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          tot[t][m] += acc[t][m];
          acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
  }
  // D[row][output]: this lane holds columns 16t + 4q .. + 3 of output 16m + r
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float* dst = part + (((size_t)bz * O + o0 + m * 16 + r) * I + i0) * C1 + t * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(dst) = tot[t][m] * inv_c;
    }
}

// ==========================================================================================================
// weight gradients on 32 x 32 tiles (v_mfma_f32_32x32x16_f16).  A matrix instruction of this shape holds the vector
// issue port for 8 of its 32 cycles (the 16 x 16 x 32 one for 8 of its 16), and a lone wave keeps the pipe full with it
// (16 x 16 x 32 from one wave: 22 to 27 cycles per instruction).
//   wave  = FE features x C1 columns = NT row tiles of 32, x 64 outputs (2 column tiles): 2 NT accumulator tiles of 16
//           registers; block = 4 waves = 4 FE features.  Grid 10: FE = 16 (224 columns, 7 row tiles, 224 accumulator
//           registers, layers whose inputs come in multiples of 64); grid 20: FE = 8 (192 columns, 6 row tiles, 192).
//   step  = PTS = 64 / FE points = KS = PTS / 4 k-steps of 16 (point, jet).  Lane (p = lane % PTS, f = lane / PTS)
//           expands its (point, feature) pair, cuts the 20 scaled jets and parks them in the wave's E tile; only the
//           4 columns written one step earlier are cleared.  Tile image: [2 planes][PTS / 2 point pairs][FE features]
//           [C1 + 6 rows][2 points x 4 jets x fp16 = 16 bytes]; the rows of a feature: 3 that take what would be
//           columns -3 .. -1, the C1 - 1 spline columns, 3 that take the columns past them, the base-activation
//           column - a pair's 4 live columns go to 4 consecutive rows whatever its interval, with no range checks,
//           and the rows in between are never read.  Lane (r = lane & 31, h = lane >> 5) reads its A operand of
//           (row tile t, k-step ks) - column 32t + r, points 4ks + 2h, + 1 - as one 16-byte word.
//   B     = gY rows of the step's points, outputs 32n + r, from global straight into registers two steps ahead, cut
//           into planes one step ahead.
// There is no second register set for long sums (the accumulators take most of the file): every kWwFlushPoints
// points the accumulators go to the chunk's partial in memory and start again from zero - the first time as plain
// stores, after that as load, add, store by the lane that owns the word (a partial of a few tens of KB per wave stays
// in the last-level cache; float atomic adds were measured at twice the time of the whole kernel).
// A wave runs in the variant that matches the knot rows of its group of 32 features (the unit in which
// bwd_weight_mfma_kernel, launched beside with skip_uniform = 1, leaves blocks out).  Output: part[chunk][O][I][C1].
// ==========================================================================================================
constexpr int kWwFlushPoints = 512;

struct GyTiles32 { u32x4 h[2], l[2]; };      // 2 output tiles x (2 points x 4 jets)

template <int C1, int FE>
struct WgW {
  static constexpr int COLS = FE * C1, NT = COLS / 32, nk = C1 + 3;
  static constexpr int PTS = 64 / FE, KS = PTS / 4, NR = NT * KS;      // points per step, k-steps, regions per step
  static constexpr int FROWS = C1 + 6;                     // rows per feature
  static constexpr int HALF = FE * FROWS * 4;              // words: one pair of points of every row
  static constexpr int PLANE = (PTS / 2) * HALF;
  static constexpr int TILE = 2 * PLANE;
  static_assert(COLS % 32 == 0 && (FE == 16 || FE == 8), "whole row tiles of 32 columns; 4 or 8 points per step");
  static constexpr int wave_words(bool gen) { return TILE + FE * (gen ? kKnotPadN : kKnotPadU); }
  // row of expanded column col = f * C1 + c of the wave (c = C1 - 1: base column)
  static __host__ __device__ constexpr int row_of(int col) {
    return (col / C1) * FROWS + ((col % C1) < C1 - 1 ? 3 + (col % C1) : FROWS - 1);
  }
};

// clear the rows a pair wrote one step earlier (at prev), write the 4 live columns c0 .. c0+3 and the base column
template <int C1, int FE>
__device__ __forceinline__ void park_wgw(unsigned* feat, unsigned*& prev, const Planes16 (&pp)[5], int c0) {
  constexpr int PL = WgW<C1, FE>::PLANE;
  const u32x2 zero = {0u, 0u};
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    *reinterpret_cast<u32x2*>(prev + m * 4) = zero;
    *reinterpret_cast<u32x2*>(prev + m * 4 + PL) = zero;
  }
  unsigned* dst = feat + (c0 + 3) * 4;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    *reinterpret_cast<u32x2*>(dst + m * 4) = pp[m].h;
    *reinterpret_cast<u32x2*>(dst + m * 4 + PL) = pp[m].l;
  }
  *reinterpret_cast<u32x2*>(feat + (WgW<C1, FE>::FROWS - 1) * 4) = pp[4].h;
  *reinterpret_cast<u32x2*>(feat + (WgW<C1, FE>::FROWS - 1) * 4 + PL) = pp[4].l;
  prev = dst;
}

template <int C1, int FE>
__device__ __forceinline__ TileOps16 read_wgw_tile(const unsigned* sE, int off) {
  TileOps16 o;
  o.h = *reinterpret_cast<const u32x4*>(sE + off);
  o.l = *reinterpret_cast<const u32x4*>(sE + off + WgW<C1, FE>::PLANE);
  return o;
}

__device__ __forceinline__ void split_gy_tile32(const f32x4 (&g)[2], const f32x4& t, GyTiles32& G, int n) {
  const Planes16 p0 = split_jet16(g[0] * t), p1 = split_jet16(g[1] * t);
  G.h[n] = (u32x4){p0.h.x, p0.h.y, p1.h.x, p1.h.y};
  G.l[n] = (u32x4){p0.l.x, p0.l.y, p1.l.x, p1.l.y};
}

__device__ __forceinline__ void mfma_wgw_tile(f32x16 (&acc)[2], const TileOps16& e, const GyTiles32& G) {
  const f16x8 eh = __builtin_bit_cast(f16x8, e.h), el = __builtin_bit_cast(f16x8, e.l);
#pragma unroll
  for (int n = 0; n < 2; ++n)
    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(el, __builtin_bit_cast(f16x8, G.h[n]), acc[n], 0, 0, 0);
#pragma unroll
  for (int n = 0; n < 2; ++n)
    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh, __builtin_bit_cast(f16x8, G.l[n]), acc[n], 0, 0, 0);
#pragma unroll
  for (int n = 0; n < 2; ++n)
    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh, __builtin_bit_cast(f16x8, G.h[n]), acc[n], 0, 0, 0);
}

// The knot reads of a pair's basis evaluation, issued ahead of the arithmetic (a step early: on one wave per SIMD an
// LDS read the next instruction depends on is a wait nothing hides).  Same interval guess and the same formulas as
// eval_basis_uniform_by_quotient / eval_basis_near_uniform.
template <bool GEN> struct KnotPre;
template <> struct KnotPre<false> { int s; float k[1]; };
template <> struct KnotPre<true> { int s; float k[8]; };

template <bool GEN>
__device__ __forceinline__ KnotPre<GEN> fetch_knots(const float* knots, int nk, const FeatGrid& fg, float x) {
  KnotPre<GEN> r;
  const float u = (x - fg.g0) * fg.inv_h;
  r.s = (u > 0.0f) ? (int)fminf(u, (float)(nk - 2)) : 0;
  if constexpr (GEN) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r.k[i] = knots[r.s - 3 + i];
  } else {
    r.k[0] = knots[r.s];
  }
  return r;
}

template <bool GEN>
__device__ __forceinline__ void basis_from_knots(const KnotPre<GEN>& kp, int nk, const FeatGrid& fg, float x, Basis& out) {
  out.inside = (int)(x >= fg.g0) & (int)(x < fg.glast);
  if constexpr (!GEN) {
    out.c0 = kp.s - kOrder;
    const float t = out.inside ? (x - kp.k[0]) * fg.inv_h : 0.0f;
    uniform_cubic_pieces<false>(t, out.inside, fg.inv_h, out);
  } else {
    const float a0 = kp.k[0], a1 = kp.k[1], a2 = kp.k[2], a3 = kp.k[3], a4 = kp.k[4], a5 = kp.k[5], a6 = kp.k[6], a7 = kp.k[7];
    const bool down = (kp.s > 0) & (x < a3), up = (kp.s < nk - 2) & (x >= a4);
    const float km2 = down ? a0 : (up ? a2 : a1);
    const float km1 = down ? a1 : (up ? a3 : a2);
    const float k0 = down ? a2 : (up ? a4 : a3);
    const float k1 = down ? a3 : (up ? a5 : a4);
    const float k2 = down ? a4 : (up ? a6 : a5);
    const float k3 = down ? a5 : (up ? a7 : a6);
    out.c0 = kp.s + (int)up - (int)down - kOrder;
    const float r1 = kan_rcp(k1 - k0);
    const float r2a = kan_rcp(k1 - km1), r2b = kan_rcp(k2 - k0);
    const float r3a = kan_rcp(k1 - km2), r3b = kan_rcp(k2 - km1), r3c = kan_rcp(k3 - k0);
    const float b1a = (k1 - x) * r1, b1b = (x - k0) * r1;
    const float u_a = b1a * r2a, u_b = b1b * r2b;
    const float b2a = (k1 - x) * u_a, b2b = (x - km1) * u_a + (k2 - x) * u_b, b2c = (x - k0) * u_b;
    const float q_a = b2a * r3a, q_b = b2b * r3b, q_c = b2c * r3c;
    const float w_a = -2.0f * u_a * r3a, w_b = 2.0f * (u_a - u_b) * r3b, w_c = 2.0f * u_b * r3c;
    const bool in = out.inside != 0;
    out.b[0] = in ? (k1 - x) * q_a : 0.0f;
    out.b[1] = in ? (x - km2) * q_a + (k2 - x) * q_b : 0.0f;
    out.b[2] = in ? (x - km1) * q_b + (k3 - x) * q_c : 0.0f;
    out.b[3] = in ? (x - k0) * q_c : 0.0f;
    out.d1[0] = in ? -3.0f * q_a : 0.0f;
    out.d1[1] = in ? 3.0f * (q_a - q_b) : 0.0f;
    out.d1[2] = in ? 3.0f * (q_b - q_c) : 0.0f;
    out.d1[3] = in ? 3.0f * q_c : 0.0f;
    out.d2[0] = in ? -3.0f * w_a : 0.0f;
    out.d2[1] = in ? 3.0f * (w_a - w_b) : 0.0f;
    out.d2[2] = in ? 3.0f * (w_b - w_c) : 0.0f;
    out.d2[3] = in ? 3.0f * w_c : 0.0f;
  }
}

// column m of a pair (m < 4: live spline column, 4: base-activation column), scaled per jet
struct JetCoefs { float s0, a1, a2, a3, q3; };
__device__ __forceinline__ JetCoefs jet_coefs(const f32x4& jet, const JetScales& sc) {
  return {sc.s[0], jet[1] * sc.s[1], jet[2] * sc.s[2], jet[3] * sc.s[3], jet[1] * jet[1] * sc.s[3]};
}
__device__ __forceinline__ f32x4 pair_column(float f0, float f1, float f2, const JetCoefs& jc) {
  return (f32x4){f0 * jc.s0, f1 * jc.a1, f1 * jc.a2, f2 * jc.q3 + f1 * jc.a3};
}

// One step = PTS points = NR regions of 6 matrix instructions, region q = (row tile q / KS, k-step q % KS), on the
// wave's E tile and the planes G.  On entry xr / gr hold the raw rows of step s+1, kp the knots around its pairs,
// cur / nxt the operands of this step's first two regions; the rows of step s+3 are fetched into xr / gr once they
// have been consumed, and the knots of step s+2 (raw rows xr2) are requested at the end.  Operands are read two
// regions ahead:
//   region 0      basis of the pairs of step s+1
//   region 1      activation, columns 0, 1
//   region 2      columns 2 .. 4, cut columns 0, 1; fetch the input jets of step s+3
//   region 3      cut columns 2 .. 4
//   region NR-3   every operand of this step has been requested: park the planes of step s+1; the reads of the
//                 last two regions go to step s+1
//   regions G0 .. scale and cut the gY rows of step s+1, one (k-step, output tile) each (G0 = NR-2, or 4 with two
//                 k-steps); the last of them fetches the gY rows of step s+3
//   region NR-1   knots of step s+2
template <int C1, int FE, int ACT, bool GEN, int NT = WgW<C1, FE>::NT, int KS = WgW<C1, FE>::KS>
__device__ __forceinline__ void wgw_step(f32x16 (&acc)[NT][2], const GyTiles32 (&G)[KS],
                                         GyTiles32 (&Gn)[KS], TileOps16& cur, TileOps16& nxt,
                                         f32x4 (&gr)[KS][2][2], f32x4& xr, KnotPre<GEN>& kp, const f32x4& xr2,
                                         KnotPre<GEN>& kp2, unsigned* sE, const int (&eoff)[NT], unsigned* feat,
                                         unsigned*& prev, const float* knots, const FeatGrid& fg, const JetScales& se,
                                         const f32x4& tg, const char* __restrict__ xrow3, const char* __restrict__ grow3,
                                         unsigned xoff, unsigned goff, int O) {
  using W = WgW<C1, FE>;
  constexpr int NR = W::NR, nk = W::nk;
  constexpr int G0 = (KS == 1) ? NR - 2 : 4, NPIECE = 2 * KS;
  static_assert(NR >= 7 && G0 + NPIECE <= NR, "the region plan below");
  Basis bs;
  Act av;
  JetCoefs jc;
  f32x4 col[5];
  Planes16 pp[5];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int t = q / KS, ks = q % KS;
    const int q2 = (q + 2) % NR;                           // (the last two regions: the tile parked in region NR-3)
    const TileOps16 nx2 = read_wgw_tile<C1, FE>(sE, eoff[q2 / KS] + (q2 % KS) * 2 * W::HALF);
    if (q == 0) basis_from_knots<GEN>(kp, nk, fg, xr[0], bs);
    if (q == 1) {
      av = eval_act(ACT, xr[0]);
      jc = jet_coefs(xr, se);
      col[0] = pair_column(bs.b[0], bs.d1[0], bs.d2[0], jc);
      col[1] = pair_column(bs.b[1], bs.d1[1], bs.d2[1], jc);
    }
    if (q == 2) {
      col[2] = pair_column(bs.b[2], bs.d1[2], bs.d2[2], jc);
      col[3] = pair_column(bs.b[3], bs.d1[3], bs.d2[3], jc);
      col[4] = pair_column(av.f0, av.f1, av.f2, jc);
      xr = ld16v(xrow3, xoff);
      pp[0] = split_jet16(col[0]); pp[1] = split_jet16(col[1]);
    }
    if (q == 3) { pp[2] = split_jet16(col[2]); pp[3] = split_jet16(col[3]); pp[4] = split_jet16(col[4]); }
    if (q == NR - 3) park_wgw<C1, FE>(feat, prev, pp, bs.c0);
    if (q >= G0 && q < G0 + NPIECE) {
      const int piece = q - G0, pk = piece / 2, pn = piece % 2;
      split_gy_tile32(gr[pk][pn], tg, Gn[pk], pn);
      if (piece == NPIECE - 1) {
#pragma unroll
        for (int k2 = 0; k2 < KS; ++k2)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 2; ++e)
              gr[k2][n][e] = ld16v(grow3, goff + (unsigned)((4 * k2 + e) * O + n * 32) * 16u);
      }
    }
    if (q == NR - 1) kp2 = fetch_knots<GEN>(knots, nk, fg, xr2[0]);
    mfma_wgw_tile(acc[t], cur, G[ks]);
    cur = nxt;
    nxt = nx2;
    interleave<6, 6>();
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the accumulators of a wave -> the chunk's partial (first: plain stores; later: load, add, store), then zero
template <int C1, int NT>
__device__ __forceinline__ void flush_wgw(f32x16 (&acc)[NT][2], float* dst0, int I, float inv_c, bool first) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      float* dst = dst0 + ((size_t)n * 32 * I) * C1 + t * 32;
      if (first) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[t][n][4 * q], acc[t][n][4 * q + 1], acc[t][n][4 * q + 2], acc[t][n][4 * q + 3]};
          *reinterpret_cast<f32x4*>(dst + 8 * q) = v * inv_c;
        }
      } else {
        f32x4 old[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) old[q] = *reinterpret_cast<const f32x4*>(dst + 8 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[t][n][4 * q], acc[t][n][4 * q + 1], acc[t][n][4 * q + 2], acc[t][n][4 * q + 3]};
          *reinterpret_cast<f32x4*>(dst + 8 * q) = old[q] + v * inv_c;
        }
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[t][n][v] = 0.f;
    }
}

template <int C1, int FE, int ACT, bool GEN>
__global__ void __launch_bounds__(256, 1)
bwd_weight_f16w_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ gy, const float* __restrict__ grid,
                       const float* __restrict__ ginfo, float* __restrict__ part, int64_t N, int I, int O,
                       int64_t chunk_pts, int nx, int ny, int nz, const unsigned* __restrict__ act_in,
                       const unsigned* __restrict__ cot_in) {
  using W = WgW<C1, FE>;
  constexpr int NT = W::NT, KS = W::KS, PTS = W::PTS, nk = W::nk;
  constexpr int KP = GEN ? kKnotPadN : kKnotPadU;
  constexpr int kWaveWords = W::wave_words(GEN);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bx, by, bz;                             // blocks that read the same rows are 8 ids apart: one XCD's L2
  {
    const int b = blockIdx.x;
    if (nz % 8 == 0) {
      const int rr = b >> 3;
      bx = rr % nx; by = (rr / nx) % ny; bz = (rr / (nx * ny)) * 8 + (b & 7);
    } else {
      bx = b % nx; by = (b / nx) % ny; bz = b / (nx * ny);
    }
  }
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int i0 = (bx * 4 + wave) * FE, o0 = by * 64;
  {                                           // the wave's group of SEL features: all equally spaced -> GEN = false
    constexpr int SEL = (FE == 16) ? 32 : 16;              // takes it; all at least nearly so -> GEN = true
    const float gi = ginfo[(i0 & ~(SEL - 1)) + (lane & (SEL - 1))];
    const bool all_near = __builtin_amdgcn_ballot_w64(gi != 0.0f) == ~0ull;
    const bool all_equal = __builtin_amdgcn_ballot_w64(gi > 0.0f) == ~0ull;
    if (!all_near || all_equal == GEN) return;
  }
  unsigned* sE = reinterpret_cast<unsigned*>(smem) + wave * kWaveWords;     // this wave's tile
  float* sK = reinterpret_cast<float*>(sE + W::TILE);                       // and its FE knot rows
  const int r = lane & 31, hq = lane >> 5, p_local = lane % PTS, f_local = lane / PTS;
  if constexpr (GEN) {                        // [g_0 x3][the nk knots][g_last x4]
    for (int idx = lane; idx < FE * KP; idx += 64) {
      const int f = idx / KP, k = idx - f * KP - 3;
      sK[idx] = grid[(size_t)(i0 + f) * nk + (k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k))];
    }
  } else {
    for (int idx = lane; idx < FE * (nk + 1); idx += 64) {
      const int f = idx / (nk + 1), k = idx - f * (nk + 1);
      sK[f * KP + k] = (k < nk) ? grid[(size_t)(i0 + f) * nk + k] : ginfo[i0 + f];
    }
  }
  for (int idx = lane; idx < W::TILE / 4; idx += 64) reinterpret_cast<u32x4*>(sE)[idx] = (u32x4){0u, 0u, 0u, 0u};
  const float* knots = sK + f_local * KP + (GEN ? 3 : 0);
  const FeatGrid fg = {grid[(size_t)(i0 + f_local) * nk], grid[(size_t)(i0 + f_local) * nk + nk - 1],
                       fabsf(ginfo[i0 + f_local])};
  // scales: gY's jet j by t_j; E's jet j by c / t_j, c = the smallest of (scale that E_j alone could take) * t_j
  JetScales se;
  f32x4 tg;
  float inv_c;
  {
    const JetScales eb = expanded_jet_scales(act_in, max_inv_h(ginfo, I, lane));
    float c = 3.0e38f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tg[j] = pow2_scale(__uint_as_float(cot_in[j]));
      c = fminf(c, eb.s[j] * tg[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) se.s[j] = c * rcp_pow2(tg[j]);
    inv_c = rcp_pow2(c);
  }

  f32x16 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[t][n][v] = 0.f;

  // N and chunk_pts are multiples of 16: whole pairs of steps
  const int64_t pbeg = (int64_t)bz * chunk_pts;
  const int64_t pend = (pbeg + chunk_pts < N) ? pbeg + chunk_pts : N;
  const char* xb = reinterpret_cast<const char*>(a);
  const char* gb = reinterpret_cast<const char*>(gy);
  const unsigned xoff = (unsigned)(p_local * I + i0 + f_local) * 16u;
  const unsigned goff = (unsigned)(2 * hq * O + o0 + r) * 16u;
  const size_t xstride = (size_t)I * 16, gstride = (size_t)O * 16;         // bytes per point
  // D[column][output]: register v of this lane is column 32t + 8 (v >> 2) + 4 hq + (v & 3) of output 32n + r
  float* const pdst = part + (((size_t)bz * O + o0 + r) * I + i0) * C1 + 4 * hq;

  if (pbeg >= pend) {                         // a chunk slot past the last point: zeros
    flush_wgw<C1, NT>(acc, pdst, I, inv_c, true);
    return;
  }

  // where this lane parks its pair (row 0 of its feature) and where it reads its operands (row tile t, k-step 0)
  unsigned* feat = sE + (p_local >> 1) * W::HALF + (f_local * W::FROWS) * 4 + (p_local & 1) * 2;
  int eoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) eoff[t] = hq * W::HALF + W::row_of(32 * t + r) * 4;
  {
    f32x4 grA[KS][2][2], grB[KS][2][2], xrA, xrB;   // raw rows of the steps after this one: A = odd steps, B = even steps
    KnotPre<GEN> kpA, kpB;
    GyTiles32 GA[KS], GB[KS];
    unsigned* prev = feat;                    // the rows the pair wrote last
    // rows past the end of the chunk: any valid rows, fetched (and expanded) but never used
    auto row_at = [&](int64_t p) { return (p < pend) ? p : pbeg; };
    {
      f32x4 g0[KS][2][2];
#pragma unroll
      for (int k2 = 0; k2 < KS; ++k2)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const unsigned off = goff + (unsigned)((4 * k2 + e) * O + n * 32) * 16u;
            g0[k2][n][e] = ld16v(gb + pbeg * gstride, off);
            grA[k2][n][e] = ld16v(gb + row_at(pbeg + PTS) * gstride, off);
            grB[k2][n][e] = ld16v(gb + row_at(pbeg + 2 * PTS) * gstride, off);
          }
      const f32x4 x0 = ld16v(xb + pbeg * xstride, xoff);
      xrA = ld16v(xb + row_at(pbeg + PTS) * xstride, xoff);
      xrB = ld16v(xb + row_at(pbeg + 2 * PTS) * xstride, xoff);
      const KnotPre<GEN> kp0 = fetch_knots<GEN>(knots, nk, fg, x0[0]);
      Basis bs;
      basis_from_knots<GEN>(kp0, nk, fg, x0[0], bs);
      const Act av = eval_act(ACT, x0[0]);
      const JetCoefs jc = jet_coefs(x0, se);
      Planes16 pp[5];
#pragma unroll
      for (int m = 0; m < 4; ++m) pp[m] = split_jet16(pair_column(bs.b[m], bs.d1[m], bs.d2[m], jc));
      pp[4] = split_jet16(pair_column(av.f0, av.f1, av.f2, jc));
      park_wgw<C1, FE>(feat, prev, pp, bs.c0);
#pragma unroll
      for (int k2 = 0; k2 < KS; ++k2) {
        split_gy_tile32(g0[k2][0], tg, GA[k2], 0);
        split_gy_tile32(g0[k2][1], tg, GA[k2], 1);
      }
      kpA = fetch_knots<GEN>(knots, nk, fg, xrA[0]);
    }
    TileOps16 cur = read_wgw_tile<C1, FE>(sE, eoff[0]);
    TileOps16 nxt = read_wgw_tile<C1, FE>(sE, (KS == 1) ? eoff[1] : eoff[0] + 2 * W::HALF);
    bool first = true;
    for (int64_t pb = pbeg; pb < pend;) {
      const int64_t pstop = (pb + kWwFlushPoints < pend) ? pb + kWwFlushPoints : pend;
      for (; pb < pstop; pb += 2 * PTS) {
        // even step (planes GA): expands step +1 (xrA, grA, kpA), fetches step +3 into xrA / grA, the knots of step +2
        wgw_step<C1, FE, ACT, GEN>(acc, GA, GB, cur, nxt, grA, xrA, kpA, xrB, kpB, sE, eoff, feat, prev, knots, fg, se, tg,
                                   xb + row_at(pb + 3 * PTS) * xstride, gb + row_at(pb + 3 * PTS) * gstride, xoff, goff, O);
        // odd step (planes GB): expands step +2 (xrB, grB, kpB), fetches step +4 into xrB / grB, the knots of step +3
        wgw_step<C1, FE, ACT, GEN>(acc, GB, GA, cur, nxt, grB, xrB, kpB, xrA, kpA, sE, eoff, feat, prev, knots, fg, se, tg,
                                   xb + row_at(pb + 4 * PTS) * xstride, gb + row_at(pb + 4 * PTS) * gstride, xoff, goff, O);
      }
      flush_wgw<C1, NT>(acc, pdst, I, inv_c, first);
      first = false;
    }
  }
}

// ==========================================================================================================
// 12 jets per scalar (5 assets + time: value, 6 first derivatives, 5 second derivatives; the basket model).
// A scalar's jets are three groups of four slots, and every (point, group) is a "virtual point" with four rows:
//   group 0: value, d/dx0, d/dx1, d/dx2      group 1: d/dx3, d/dx4, d/dx5, d2/dx0^2      group 2: d2/dx1^2 .. d2/dx4^2
// Row r of a group is  al_r * B + be_r * B' + ga_r * B''  of the live bases (and of the activation), with
//   value: (1, 0, 0)      d/dx_k: (0, a[1+k], 0)      d2/dx_k^2: (0, a[7+k], a[1+k]^2)
// (kan_math.h, jet_accumulate), each multiplied by its slot's scale.  Forward and input gradients take 8 points x 3
// groups = 6 row tiles per wave; the weight gradients walk the virtual points of their chunk 8 at a time.
// ==========================================================================================================
constexpr int kRowWords12 = 56;              // 8 points x 12 slots x fp16 = 192 bytes + 32 bytes of padding; 8 * 7: conflict-free as above

struct SlotScales12 { float s[12]; };

// Upper bounds of the expanded rows from the per-slot maxima of the input jets (12 words) -> one scale per slot
__device__ __forceinline__ SlotScales12 expanded_slot_scales12(const unsigned* amax, float inv_h_max) {
  float a[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) a[j] = __uint_as_float(amax[j]);
  const float c1 = fmaxf(1.1f, 1.5f * inv_h_max), c2 = fmaxf(0.5f, 3.0f * inv_h_max * inv_h_max);
  SlotScales12 r;
  r.s[0] = pow2_scale(fmaxf(1.0f, a[0]));
#pragma unroll
  for (int k = 0; k < 6; ++k) r.s[1 + k] = pow2_scale(c1 * a[1 + k]);
#pragma unroll
  for (int k = 0; k < 5; ++k) r.s[7 + k] = pow2_scale(c2 * a[1 + k] * a[1 + k] + c1 * a[7 + k]);
  return r;
}

// out[j] = max(out[j], bits of max |t[e][j]|), j < 12, over e < n: per-slot maxima of a 12-jet tensor [...][12]
__global__ void jet_absmax12_kernel(const f32x4* __restrict__ t, int64_t n, unsigned* __restrict__ out) {
  f32x4 m[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const f32x4 v0 = t[3 * e], v1 = t[3 * e + 1], v2 = t[3 * e + 2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[0][j] = fmaxf(m[0][j], fabsf(v0[j])); m[1][j] = fmaxf(m[1][j], fabsf(v1[j])); m[2][j] = fmaxf(m[2][j], fabsf(v2[j]));
    }
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) publish_jet_max(m[g], out + 4 * g, threadIdx.x & 63);
}

struct PairCols12 { f32x4 col[5][3]; int c0; };            // [4 live columns + base column][group] x 4 rows
struct Planes12 { u32x2 h[3], l[3]; };                     // one column: 3 groups x 4 rows, two planes

// row coefficients of the three groups of one input jet, scaled: be[g][r] multiplies B', ga[g][r] multiplies B''
// (group 0 row 0 is the value row: s[0] * B)
struct RowCoefs12 { f32x4 be[3]; f32x4 ga[3]; float s0; };

__device__ __forceinline__ RowCoefs12 row_coefs12(const f32x4 (&x)[3], const SlotScales12& sc) {
  RowCoefs12 rc;
  const float a1 = x[0][1], a2 = x[0][2], a3 = x[0][3], a4 = x[1][0], a5 = x[1][1];
  rc.s0 = sc.s[0];
  rc.be[0] = (f32x4){0.f, a1 * sc.s[1], a2 * sc.s[2], a3 * sc.s[3]};
  rc.be[1] = (f32x4){a4 * sc.s[4], a5 * sc.s[5], x[1][2] * sc.s[6], x[1][3] * sc.s[7]};
  rc.be[2] = (f32x4){x[2][0] * sc.s[8], x[2][1] * sc.s[9], x[2][2] * sc.s[10], x[2][3] * sc.s[11]};
  rc.ga[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  rc.ga[1] = (f32x4){0.f, 0.f, 0.f, a1 * a1 * sc.s[7]};
  rc.ga[2] = (f32x4){a2 * a2 * sc.s[8], a3 * a3 * sc.s[9], a4 * a4 * sc.s[10], a5 * a5 * sc.s[11]};
  return rc;
}

template <int ACT, bool GEN>
__device__ __forceinline__ void expand12(const float* knots, int nk, const FeatGrid& fg, const f32x4 (&x)[3],
                                         const SlotScales12& sc, PairCols12& out) {
  Basis bs;
  if constexpr (GEN) eval_basis_near_uniform(knots, nk, fg.g0, fg.glast, fg.inv_h, x[0][0], bs);
  else eval_basis_uniform_by_quotient(knots, nk, fg.g0, fg.glast, fg.inv_h, x[0][0], bs);
  const Act av = eval_act(ACT, x[0][0]);
  const RowCoefs12 rc = row_coefs12(x, sc);
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const float b0 = m < 4 ? bs.b[m & 3] : av.f0, b1 = m < 4 ? bs.d1[m & 3] : av.f1, b2 = m < 4 ? bs.d2[m & 3] : av.f2;
    out.col[m][0] = rc.be[0] * b1;
    out.col[m][0][0] = rc.s0 * b0;
    out.col[m][1] = rc.be[1] * b1;
    out.col[m][1][3] += rc.ga[1][3] * b2;
    out.col[m][2] = rc.be[2] * b1 + rc.ga[2] * b2;
  }
  out.c0 = bs.c0;
}

struct PairKeep12x { Basis bs; Act av; };                   // basis and activation of a pair, kept over several regions

// one expanded column (m = 0..3 live spline columns, 4 = base column) of all three groups, from the pieces above
__device__ __forceinline__ void column12(const PairKeep12x& pk, const RowCoefs12& rc, int m, f32x4 (&col)[3]) {
  const float b0 = m < 4 ? pk.bs.b[m & 3] : pk.av.f0, b1 = m < 4 ? pk.bs.d1[m & 3] : pk.av.f1;
  const float b2 = m < 4 ? pk.bs.d2[m & 3] : pk.av.f2;
  col[0] = rc.be[0] * b1;
  col[0][0] = rc.s0 * b0;
  col[1] = rc.be[1] * b1;
  col[1][3] += rc.ga[1][3] * b2;
  col[2] = rc.be[2] * b1 + rc.ga[2] * b2;
}

// the same for the column a half-wave takes when the two halves of a wave, which hold the same pairs, share a pair's
// columns: step j = 0, 1: live column j (lanes 0..31) or 2 + j (lanes 32..63); j = 2: the base column (upper half: unused)
__device__ __forceinline__ void column12_half(const PairKeep12x& pk, const RowCoefs12& rc, int j, bool upper, f32x4 (&col)[3]) {
  float b0, b1, b2;
  if (j < 2) {
    b0 = upper ? pk.bs.b[2 + (j & 1)] : pk.bs.b[j & 1];
    b1 = upper ? pk.bs.d1[2 + (j & 1)] : pk.bs.d1[j & 1];
    b2 = upper ? pk.bs.d2[2 + (j & 1)] : pk.bs.d2[j & 1];
  } else {
    b0 = pk.av.f0; b1 = pk.av.f1; b2 = pk.av.f2;
  }
  col[0] = rc.be[0] * b1;
  col[0][0] = rc.s0 * b0;
  col[1] = rc.be[1] * b1;
  col[1][3] += rc.ga[1][3] * b2;
  col[2] = rc.be[2] * b1 + rc.ga[2] * b2;
}

__device__ __forceinline__ Planes12 split_col12(const f32x4 (&c)[3]) {
  Planes12 p;
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const Planes16 q = split_jet16(c[g]);
    p.h[g] = q.h;
    p.l[g] = q.l;
  }
  return p;
}

// E tile row (one expanded column): [group][8 points][4 rows] fp16 = 3 x 16 words (+ pad); row tile 2g + (p >> 2)
// of the A operand is the 8 words at 16g + 8(p >> 2)
template <int C1, int PLANE>
__device__ __forceinline__ void park_pair12(unsigned* base, unsigned* dump, const Planes12 (&pp)[5], int c0, int& prev_c0) {
  constexpr int C = C1 - 1;
  const u32x2 zero = {0u, 0u};
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = prev_c0 + m;
    unsigned* dst = (c >= 0 && c < C) ? base + c * kRowWords12 : dump;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      *reinterpret_cast<u32x2*>(dst + 16 * g) = zero;
      *reinterpret_cast<u32x2*>(dst + 16 * g + PLANE) = zero;
    }
  }
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const int c = (m < 4) ? c0 + m : C;
    unsigned* dst = (m == 4 || (c >= 0 && c < C)) ? base + c * kRowWords12 : dump;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      *reinterpret_cast<u32x2*>(dst + 16 * g) = pp[m].h[g];
      *reinterpret_cast<u32x2*>(dst + 16 * g + PLANE) = pp[m].l[g];
    }
  }
  prev_c0 = c0;
}

// the park of a half-wave (column12_half): clears the two live columns this half wrote a stage earlier, writes its two
// live columns and (lower half) the base column; the upper half's third column goes to the dump row
template <int C1, int PLANE>
__device__ __forceinline__ void park_pair12_half(unsigned* base, unsigned* dump, const Planes12 (&pp)[3], int c0, int& prev_c0,
                                                 bool upper) {
  constexpr int C = C1 - 1;
  const u32x2 zero = {0u, 0u};
  const int h2 = upper ? 2 : 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = prev_c0 + h2 + j;
    unsigned* dst = (c >= 0 && c < C) ? base + c * kRowWords12 : dump;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      *reinterpret_cast<u32x2*>(dst + 16 * g) = zero;
      *reinterpret_cast<u32x2*>(dst + 16 * g + PLANE) = zero;
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = c0 + h2 + j;
    unsigned* dst = (j == 2) ? (upper ? dump : base + C * kRowWords12) : ((c >= 0 && c < C) ? base + c * kRowWords12 : dump);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      *reinterpret_cast<u32x2*>(dst + 16 * g) = pp[j].h[g];
      *reinterpret_cast<u32x2*>(dst + 16 * g + PLANE) = pp[j].l[g];
    }
  }
  prev_c0 = c0;
}

// ---- forward, 12 jets: wave = 8 points x 3 groups = 6 row tiles x 64 outputs (24 accumulator tiles), one wave per
// SIMD.  Stage = 4 features = KS k-steps x 6 row tiles = 6*KS regions of 12 matrix instructions.  Lanes 32..63
// hold the pairs of lanes 0..31 again: the two halves share a pair's columns (two live columns + the base column
// below, the other two live columns above).  N must be a multiple of 8.
template <int C1, int ACT, bool GEN>
struct Fwd12 {
  static constexpr int CP = (C1 + 7) / 8 * 8;
  static constexpr int KS = CP / 8;
  static constexpr int NR = 6 * KS;
  static constexpr int ROWS = 4 * C1 + 2;          // + a row of zeros (padding columns) + a dump row (columns off the grid)
  static constexpr int PLANE = ROWS * kRowWords12;
  static constexpr int KP = GEN ? kKnotPadN : kKnotPadU;
  static constexpr int nk = C1 + 3;
  static constexpr int kStepVecs = 4 * 2 * 64;
  static constexpr int kStageVecs = KS * kStepVecs;

  // The expansion of the next stage's pairs is spread over the regions, a few vector instructions behind each matrix
  // instruction (one wave per SIMD: whatever vector work sits between two matrix instructions is time the matrix
  // pipe idles once it exceeds two instructions):
  //   region 0            basis and activation of the pairs
  //   region 1            row coefficients of the three groups; fetch the input jets of the stage after the next
  //   regions 1 .. 3      this half-wave's column j = region - 1 of all groups
  //   regions 2 .. 4      cut column j = region - 2
  //   region NR-2         park the next stage's planes
  template <bool FLIP>
  static __device__ __forceinline__ void stage(f32x4 (&acc)[6][4], WTiles16& Wa, WTiles16& Wb, ATile16& cur, ATile16& nxt,
                                               f32x4 (&xraw)[3], unsigned* sE, const float* knots_next,
                                               const SlotScales12& sc, int& prev_c0, const u32x4* __restrict__ wsrc,
                                               const u32x4* __restrict__ wnext, const char* __restrict__ xnext2,
                                               const int (&aoff)[KS][2], int lane) {
    static_assert(NR >= 9, "the plan below needs 7 regions before the park");
    const int p_local = lane & 7, f_local = (lane >> 3) & 3;
    const bool upper = lane >= 32;
    PairKeep12x pk;
    RowCoefs12 rc;
    f32x4 col[3];
    Planes12 pp[3];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int ks = i / 6, m = i % 6;
      const bool odd = ((ks & 1) != 0) != FLIP;
      WTiles16& Wc = odd ? Wb : Wa;
      WTiles16& Wn = odd ? Wa : Wb;
      if (m < 4) load_w16(Wn, (ks + 1 < KS) ? wsrc + (ks + 1) * kStepVecs : wnext, lane, m);
      if (i >= 2 && i <= 4) pp[i - 2] = split_col12(col);           // (the column the region before computed)
      if (i == 0) {
        const FeatGrid fg = {knots_next[0], knots_next[nk - 1], fabsf(knots_next[GEN ? nk + 4 : nk])};
        if constexpr (GEN) eval_basis_near_uniform(knots_next, nk, fg.g0, fg.glast, fg.inv_h, xraw[0][0], pk.bs);
        else eval_basis_uniform_by_quotient(knots_next, nk, fg.g0, fg.glast, fg.inv_h, xraw[0][0], pk.bs);
        pk.av = eval_act(ACT, xraw[0][0]);
      }
      if (i == 1) {
        rc = row_coefs12(xraw, sc);
#pragma unroll
        for (int g = 0; g < 3; ++g) xraw[g] = *reinterpret_cast<const f32x4*>(xnext2 + 16 * g);
      }
      if (i >= 1 && i <= 3) column12_half(pk, rc, i - 1, upper, col);
      if (i == NR - 2) {
        __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): retire the reads in flight before the run of writes
        unsigned* base = sE + (f_local * C1) * kRowWords12 + p_local * 2;
        park_pair12_half<C1, PLANE>(base, sE + (ROWS - 1) * kRowWords12 + p_local * 2, pp, pk.bs.c0, prev_c0, upper);
      }
      const int t2 = (i + 2) % NR;
      const ATile16 far = read_atile16<PLANE>(sE, aoff[t2 / 6][0], aoff[t2 / 6][1], t2 % 6);
      mfma3x4(acc[m], cur.h, cur.l, Wc.h, Wc.l);
      cur = nxt;
      nxt = far;
      if (i <= 4) interleave<12, 5>(); else interleave<12, 3>();
      PDEKAN_REGION_FENCE();
    }
  }
};

template <int C1, int ACT, bool GEN>
__global__ void __launch_bounds__(256, 1)
fwd12_f16_kernel(const f32x4* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
                 const u32x4* __restrict__ wpl, f32x4* __restrict__ y, int64_t N, int I, int O,
                 const unsigned* __restrict__ amax_in, const unsigned* __restrict__ wmax) {
  using F = Fwd12<C1, ACT, GEN>;
  constexpr int KS = F::KS, KP = F::KP, nk = F::nk, CP = F::CP;
  constexpr int kWaveWords = 2 * F::PLANE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  {
    bool near = true, equal = true;
    for (int i = lane; i < ((I + 63) & ~63); i += 64) {
      const float gi = ginfo[i < I ? i : 0];
      near = near && (__builtin_amdgcn_ballot_w64(gi != 0.0f) == ~0ull);
      equal = equal && (__builtin_amdgcn_ballot_w64(gi > 0.0f) == ~0ull);
    }
    if (!near || equal == GEN) return;
  }
  float* sK = reinterpret_cast<float*>(smem);
  unsigned* sE = reinterpret_cast<unsigned*>(smem) + I * KP + wave * kWaveWords;
  if constexpr (GEN) {
    for (int idx = tid; idx < I * KP; idx += 256) {
      const int f = idx / KP, k = idx - f * KP - 3;
      sK[idx] = (k == nk + 4) ? ginfo[f] : grid[(size_t)f * nk + (k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k))];
    }
  } else {
    for (int idx = tid; idx < I * (nk + 1); idx += 256) {
      const int f = idx / (nk + 1), k = idx - f * (nk + 1);
      sK[f * KP + k] = (k < nk) ? grid[(size_t)f * nk + k] : ginfo[f];
    }
  }
  for (int idx = lane; idx < kWaveWords; idx += 64) sE[idx] = 0u;
  const SlotScales12 sc = expanded_slot_scales12(amax_in, max_inv_h(ginfo, I, lane));
  const float sw = pow2_scale(__uint_as_float(wmax[0]));
  __syncthreads();

  const int r = lane & 15, kb = lane >> 4, p_local = lane & 7, f_local = (lane >> 3) & 3;
  const int64_t P0 = ((int64_t)blockIdx.x * 4 + wave) * 8;
  if (P0 >= N) return;
  const int o0 = blockIdx.y * 64;
  const int nst = I / 4;
  int aoff[KS][2];
  {
    const int q_ = (lane & 15) >> 2, p_ = lane & 3;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int col = ks * 32 + kstep_col(kb, half * 4 + q_);
        const int row = ((col % CP) < C1) ? (col / CP) * C1 + (col % CP) : F::ROWS - 2;
        aoff[ks][half] = row * kRowWords12 + 2 * p_;
      }
  }
  const char* xrow = reinterpret_cast<const char*>(a + ((P0 + p_local) * I + f_local) * 3);
  constexpr size_t kStageBytes = 4 * 48;                                    // 4 features x 12 floats
  const u32x4* wbase = wpl + (size_t)blockIdx.y * nst * F::kStageVecs;
  const float* kbase = sK + f_local * KP + (GEN ? 3 : 0);

  f32x4 acc[6][4], tot[6][4];
#pragma unroll
  for (int m = 0; m < 6; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) { acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; tot[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  WTiles16 WA, WB;
  int prev_c0 = -100;
  f32x4 xraw[3];
  {
    f32x4 x0[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      x0[g] = *reinterpret_cast<const f32x4*>(xrow + 16 * g);
      xraw[g] = *reinterpret_cast<const f32x4*>(xrow + kStageBytes + 16 * g);
    }
    const FeatGrid fg = {kbase[0], kbase[nk - 1], fabsf(kbase[GEN ? nk + 4 : nk])};
    PairCols12 pj;
    Planes12 pp[5];
    expand12<ACT, GEN>(kbase, nk, fg, x0, sc, pj);
#pragma unroll
    for (int m = 0; m < 5; ++m) pp[m] = split_col12(pj.col[m]);
    unsigned* base = sE + (f_local * C1) * kRowWords12 + p_local * 2;
    park_pair12<C1, F::PLANE>(base, sE + (F::ROWS - 1) * kRowWords12 + p_local * 2, pp, pj.c0, prev_c0);
#pragma unroll
    for (int n = 0; n < 4; ++n) load_w16(WA, wbase, lane, n);
  }
  ATile16 cur = read_atile16<F::PLANE>(sE, aoff[0][0], aoff[0][1], 0);
  ATile16 nxt = read_atile16<F::PLANE>(sE, aoff[0][0], aoff[0][1], 1);
  for (int s = 0; s < nst; s += 2) {
    const int s1 = s + 1, s2 = (s + 2 < nst) ? s + 2 : 0, s3 = (s + 3 < nst) ? s + 3 : 0;
    F::template stage<false>(acc, WA, WB, cur, nxt, xraw, sE, kbase + 4 * s1 * KP, sc, prev_c0,
                             wbase + (size_t)s * F::kStageVecs, wbase + (size_t)s1 * F::kStageVecs,
                             xrow + (size_t)s2 * kStageBytes, aoff, lane);
    F::template stage<(KS & 1) != 0>(acc, WA, WB, cur, nxt, xraw, sE, kbase + 4 * s2 * KP, sc, prev_c0,
                                     wbase + (size_t)s1 * F::kStageVecs, wbase + (size_t)s2 * F::kStageVecs,
                                     xrow + (size_t)s3 * kStageBytes, aoff, lane);
    if (((s + 2) % kFwdFlushStages) == 0) {
#pragma unroll
      for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) { tot[m][n] += acc[m][n]; acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
  }
  // D[(p, slot)][output]: row tile m = 2g + pq; this lane holds slots 4g .. 4g+3 of point 4pq + kb for output 16n + r
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    const int g = m >> 1, pq = m & 1;
    f32x4 inv;
#pragma unroll
    for (int j = 0; j < 4; ++j) inv[j] = rcp_pow2(sc.s[4 * g + j] * sw);
    const int64_t p = P0 + 4 * pq + kb;
#pragma unroll
    for (int n = 0; n < 4; ++n) y[(p * O + o0 + 16 * n + r) * 3 + g] = (acc[m][n] + tot[m][n]) * inv;
  }
}

// ---- forward, 12 jets, on 32 x 32 x 16 tiles.  The lone wave a SIMD holds here (192 accumulator registers, a 25 KB E
// tile) issues a 16 x 16 x 32 instruction every 22 cycles at best; the 32 x 32 x 16 shape keeps the pipe full from one
// wave and holds the vector issue port for 8 of its 32 cycles.  Same wave tile, E tile, expansion and park as
// fwd12_f16_kernel: a group of slots (8 points x 4 rows) is one row tile of 32, the 16 padded columns of a feature
// one k-step; 3 row tiles x 2 output tiles of 32 = 6 accumulator tiles of 16 registers.  A stage = 4 features = 12
// regions (feature, group) of 6 matrix instructions.  B operands (forward image of pack_wplanes16_kernel<C1, true>):
// one feature = 4 x 16 bytes per lane, fetched three features ahead; A operands by transposed reads two regions ahead,
// columns in wide_col order.

struct WFeat32 { u32x4 h[2], l[2]; };                       // B operands of one feature: [n]

template <int C1, int ACT, bool GEN>
struct Fwd12W {
  using F = Fwd12<C1, ACT, GEN>;
  static_assert(F::CP == 16, "one k-step of 16 columns per feature");
  static constexpr int NR = 12;
  static constexpr int kFeatVecs = 2 * 2 * 64;             // u32x4 per feature of W planes
  static constexpr int nk = F::nk;

  static __device__ __forceinline__ void load_w(WFeat32& w, const u32x4* __restrict__ src, int lane, int n) {
    w.h[n] = src[(n * 2) * 64 + lane];
    w.l[n] = src[(n * 2 + 1) * 64 + lane];
  }
  static __device__ __forceinline__ ATile16 read_a(const unsigned* sE, int off0, int off1, int g) {
    ATile16 a;
    a.h = tr_read8(sE + off0 + 16 * g, sE + off1 + 16 * g);
    a.l = tr_read8(sE + off0 + 16 * g + F::PLANE, sE + off1 + 16 * g + F::PLANE);
    return a;
  }

  // region i = (feature i / 3, group i % 3); the expansion of the next stage's pairs as in Fwd12::stage
  static __device__ __forceinline__ void stage(f32x16 (&acc)[3][2], WFeat32 (&Wr)[4], ATile16& cur, ATile16& nxt,
                                               f32x4 (&xraw)[3], KnotPre<GEN>& kp, FeatGrid& fg, unsigned* sE,
                                               const float* knots_next2,
                                               const SlotScales12& sc, int& prev_c0, const u32x4* __restrict__ wsrc,
                                               const u32x4* __restrict__ wnext, const char* __restrict__ xnext2,
                                               const int (&aoff)[4][2], int lane) {
    const int p_local = lane & 7, f_local = (lane >> 3) & 3;
    const bool upper = lane >= 32;
    PairKeep12x pk;
    RowCoefs12 rc;
    f32x4 col[3];
    Planes12 pp[3];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int f = i / 3, g = i % 3;
      const WFeat32& Wc = Wr[f];
      if (g < 2) {                             // B operands three features ahead (a ring of four: the slot feature f-1 left)
        const int t = f + 3;
        load_w(Wr[t & 3], (t < 4) ? wsrc + t * kFeatVecs : wnext + (t - 4) * kFeatVecs, lane, g);
      }
      if (i >= 2 && i <= 4) pp[i - 2] = split_col12(col);           // (the column the region before computed)
      if (i == 0) {                            // (the knots around the pair were fetched at the end of the stage before)
        basis_from_knots<GEN>(kp, nk, fg, xraw[0][0], pk.bs);
        pk.av = eval_act(ACT, xraw[0][0]);
      }
      if (i == 1) {
        rc = row_coefs12(xraw, sc);
#pragma unroll
        for (int q = 0; q < 3; ++q) xraw[q] = *reinterpret_cast<const f32x4*>(xnext2 + 16 * q);
      }
      if (i >= 1 && i <= 3) column12_half(pk, rc, i - 1, upper, col);
      if (i == NR - 2) {
        unsigned* base = sE + (f_local * C1) * kRowWords12 + p_local * 2;
        park_pair12_half<C1, F::PLANE>(base, sE + (F::ROWS - 1) * kRowWords12 + p_local * 2, pp, pk.bs.c0, prev_c0, upper);
      }
      if (i == NR - 1) {                       // knots of the pairs of the stage after the next (xraw since region 1)
        fg = (FeatGrid){knots_next2[0], knots_next2[nk - 1], fabsf(knots_next2[GEN ? nk + 4 : nk])};
        kp = fetch_knots<GEN>(knots_next2, nk, fg, xraw[0][0]);
      }
      const int t2 = (i + 2) % NR;           // (the last two regions read the next stage's first feature)
      const ATile16 far = read_a(sE, aoff[t2 / 3][0], aoff[t2 / 3][1], t2 % 3);
      const f16x8 Ah = __builtin_bit_cast(f16x8, cur.h), Al = __builtin_bit_cast(f16x8, cur.l);
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[g][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(f16x8, Wc.h[n]), acc[g][n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[g][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, Wc.l[n]), acc[g][n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[g][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, Wc.h[n]), acc[g][n], 0, 0, 0);
      cur = nxt;
      nxt = far;
      if (i <= 4) interleave<6, 10>(); else interleave<6, 6>();
      PDEKAN_REGION_FENCE();
    }
  }
};

template <int C1, int ACT, bool GEN>
__global__ void __launch_bounds__(256, 1)
fwd12w_f16_kernel(const f32x4* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
                  const u32x4* __restrict__ wpl, f32x4* __restrict__ y, int64_t N, int I, int O,
                  const unsigned* __restrict__ amax_in, const unsigned* __restrict__ wmax) {
  using F = Fwd12<C1, ACT, GEN>;
  using W = Fwd12W<C1, ACT, GEN>;
  constexpr int KP = F::KP, nk = F::nk;
  constexpr int kWaveWords = 2 * F::PLANE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  {
    bool near = true, equal = true;
    for (int i = lane; i < ((I + 63) & ~63); i += 64) {
      const float gi = ginfo[i < I ? i : 0];
      near = near && (__builtin_amdgcn_ballot_w64(gi != 0.0f) == ~0ull);
      equal = equal && (__builtin_amdgcn_ballot_w64(gi > 0.0f) == ~0ull);
    }
    if (!near || equal == GEN) return;
  }
  float* sK = reinterpret_cast<float*>(smem);
  unsigned* sE = reinterpret_cast<unsigned*>(smem) + I * KP + wave * kWaveWords;
  if constexpr (GEN) {
    for (int idx = tid; idx < I * KP; idx += 256) {
      const int f = idx / KP, k = idx - f * KP - 3;
      sK[idx] = (k == nk + 4) ? ginfo[f] : grid[(size_t)f * nk + (k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k))];
    }
  } else {
    for (int idx = tid; idx < I * (nk + 1); idx += 256) {
      const int f = idx / (nk + 1), k = idx - f * (nk + 1);
      sK[f * KP + k] = (k < nk) ? grid[(size_t)f * nk + k] : ginfo[f];
    }
  }
  for (int idx = lane; idx < kWaveWords; idx += 64) sE[idx] = 0u;
  const SlotScales12 sc = expanded_slot_scales12(amax_in, max_inv_h(ginfo, I, lane));
  const float sw = pow2_scale(__uint_as_float(wmax[0]));
  __syncthreads();

  const int r = lane & 31, hq = lane >> 5, p_local = lane & 7, f_local = (lane >> 3) & 3;
  const int64_t P0 = ((int64_t)blockIdx.x * 4 + wave) * 8;
  if (P0 >= N) return;
  const int o0 = blockIdx.y * 64;
  const int nst = I / 4;
  // E tile rows of this lane's transposed reads: lane 4q+p of 16-lane group t addresses the row of column
  // wide_col(t >> 1, 4 * half + q) of the feature, (point, row) block 16 (t & 1) + 4p .. + 3 of the group's 32
  int aoff[4][2];
  {
    const int t = lane >> 4, q_ = (lane & 15) >> 2, p_ = lane & 3;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int c = wide_col(t >> 1, 4 * half + q_);
        const int row = (c < C1) ? f * C1 + c : F::ROWS - 2;
        aoff[f][half] = row * kRowWords12 + 8 * (t & 1) + 2 * p_;
      }
  }
  const char* xrow = reinterpret_cast<const char*>(a + ((P0 + p_local) * I + f_local) * 3);
  constexpr size_t kStageBytes = 4 * 48;                                    // 4 features x 12 floats
  const u32x4* wbase = wpl + (size_t)blockIdx.y * I * W::kFeatVecs;
  const float* kbase = sK + f_local * KP + (GEN ? 3 : 0);

  f32x16 acc[3][2], tot[3][2];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int v = 0; v < 16; ++v) { acc[g][n][v] = 0.f; tot[g][n][v] = 0.f; }

  WFeat32 Wr[4];
  int prev_c0 = -100;
  f32x4 xraw[3];
  {
    f32x4 x0[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      x0[g] = *reinterpret_cast<const f32x4*>(xrow + 16 * g);
      xraw[g] = *reinterpret_cast<const f32x4*>(xrow + kStageBytes + 16 * g);
    }
    const FeatGrid fg = {kbase[0], kbase[nk - 1], fabsf(kbase[GEN ? nk + 4 : nk])};
    PairCols12 pj;
    Planes12 pp[5];
    expand12<ACT, GEN>(kbase, nk, fg, x0, sc, pj);
#pragma unroll
    for (int m = 0; m < 5; ++m) pp[m] = split_col12(pj.col[m]);
    unsigned* base = sE + (f_local * C1) * kRowWords12 + p_local * 2;
    park_pair12<C1, F::PLANE>(base, sE + (F::ROWS - 1) * kRowWords12 + p_local * 2, pp, pj.c0, prev_c0);
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      W::load_w(Wr[f], wbase + f * W::kFeatVecs, lane, 0);
      W::load_w(Wr[f], wbase + f * W::kFeatVecs, lane, 1);
    }
  }
  ATile16 cur = W::read_a(sE, aoff[0][0], aoff[0][1], 0);
  ATile16 nxt = W::read_a(sE, aoff[0][0], aoff[0][1], 1);
  FeatGrid fgn;
  KnotPre<GEN> kpn;
  {
    const float* kn = kbase + 4 * (nst > 1 ? 1 : 0) * KP;
    fgn = (FeatGrid){kn[0], kn[nk - 1], fabsf(kn[GEN ? nk + 4 : nk])};
    kpn = fetch_knots<GEN>(kn, nk, fgn, xraw[0][0]);
  }
  for (int s = 0; s < nst; ++s) {
    // stages past the end: any valid data, expanded and parked but never multiplied
    const int s1 = (s + 1 < nst) ? s + 1 : 0, s2 = (s + 2 < nst) ? s + 2 : 0;
    W::stage(acc, Wr, cur, nxt, xraw, kpn, fgn, sE, kbase + 4 * s2 * KP, sc, prev_c0, wbase + (size_t)s * 4 * W::kFeatVecs,
             wbase + (size_t)s1 * 4 * W::kFeatVecs, xrow + (size_t)s2 * kStageBytes, aoff, lane);
    if (((s + 1) % kFwdFlushStages) == 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          tot[g][n] += acc[g][n];
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[g][n][v] = 0.f;
        }
    }
  }
  // D[(p, row)][output]: registers 4q .. 4q+3 of tile (g, n) are slots 4g .. 4g+3 of point 2q + hq for output 32n + r
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    f32x4 inv;
#pragma unroll
    for (int j = 0; j < 4; ++j) inv[j] = rcp_pow2(sc.s[4 * g + j] * sw);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t p = P0 + 2 * q + hq;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const f32x16 t = acc[g][n] + tot[g][n];
        y[(p * O + o0 + 32 * n + r) * 3 + g] = (f32x4){t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]} * inv;
      }
    }
  }
}

// ---- forward, 12 jets, on the structured-sparsity matrix instruction (see fwd_sp_f16_kernel for the operand layout;
// the same image of the W planes).  A row tile is ONE slot of the wave's 16 points, so lane (p, f) of the expansion
// holds the A operands of all 12 row tiles of its pair: 12 slots x 2 planes x 4 words.  Wave = 16 points x 12 slots x
// 64 outputs = 48 accumulator tiles (192 registers), one wave per SIMD; every lane expands a pair of its own (the
// dense kernels' 8-point tiles leave half of the lanes with copies).  A stage (4 features) is 144 matrix
// instructions in 12 sub-steps of 12: slots (2u, 2u+1) against output tiles 0, 1 (u < 6), then against tiles 2, 3.
// The A operands are rewritten in place: a pair of slots is cut for the NEXT stage one sub-step after its last
// use (sub-steps 7 .. 11; slots 10, 11 in sub-step 0 of the next stage), the basis, activation and row coefficients
// of the next stage's pair are evaluated under sub-steps 1 .. 6, the B tiles 0, 1 / 2, 3 are fetched for the next
// stage as soon as their half of the stage is issued, the input jets three stages ahead.  Persistent blocks: the
// stages of a wave's tiles form one stream, as in fwd_sp_f16_kernel.  No second register set for long sums:
// layers with more than kFwdChainMax chained instructions per accumulator (I > 168) keep the dense kernel.
struct SpA12 { u32x4 h[12], l[12]; };

// one word of each plane for (x, 0); l comes in with its high half zero and keeps it
__device__ __forceinline__ void split_single16(float x, unsigned& h, unsigned& l) {
  asm("v_cvt_pk_f16_f32 %0, %1, 0" : "=v"(h) : "v"(x));
  asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x));
}

// basis and activation of a pair in group order (the live column of group g; zero where that column is not one of
// the layer's) and the positions of the live columns inside their groups
struct SpPair { float b[4], d1[4], d2[4], f0, f1, f2; };

template <int C1, int ACT, bool GEN>
__device__ __forceinline__ unsigned sp_pair(const float* knots, int nk, float x, SpPair& out) {
  constexpr int C = C1 - 1;
  static_assert(C == 13, "the base column sits in group 1, slot 3");
  const FeatGrid fg = {knots[0], knots[nk - 1], fabsf(knots[GEN ? nk + 4 : nk])};
  Basis bs;
  if constexpr (GEN) eval_basis_near_uniform(knots, nk, fg.g0, fg.glast, fg.inv_h, x, bs);
  else eval_basis_uniform_by_quotient(knots, nk, fg.g0, fg.glast, fg.inv_h, x, bs);
  const Act av = eval_act(ACT, x);
  const int r = bs.c0 & 3, base4 = bs.c0 >> 2;
  sp_rotate(bs.b, r); sp_rotate(bs.d1, r); sp_rotate(bs.d2, r);
  unsigned idx = 0xCCCCu;                                  // (see sp_expand: only group 1 needs its values zeroed)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int slot = base4 + (g < r ? 1 : 0);
    const bool ok = g != 1 || (slot >= 0 && 4 * slot + g < C);
    out.b[g] = ok ? bs.b[g] : 0.0f; out.d1[g] = ok ? bs.d1[g] : 0.0f; out.d2[g] = ok ? bs.d2[g] : 0.0f;
    idx |= (unsigned)(slot & 3) << (4 * g);
  }
  out.f0 = av.f0; out.f1 = av.f1; out.f2 = av.f2;
  return idx;
}

// The 12 row tiles of a wave are the 12 slots of the jets of 16 points.  (The 4 jets of 3 x 16 points on the same
// frame - the B operands fetched once for three times the points - take the time of fwd_sp_f16_kernel: what bounds
// both is the vector work per pair, which one wave per SIMD hides no better than two.)
template <int C1, int ACT, bool GEN>
struct SpRows12 {
  using Scales = SlotScales12;
  static constexpr int NIDX = 1;                           // row tiles j share the positions idx[j * NIDX / 12]
  static constexpr int POINTS = 16, VECS = 3;              // points per wave tile, f32x4 per (point, feature)
  struct State { SpPair pr; RowCoefs12 rc; };
  static __device__ __forceinline__ Scales scales(const unsigned* amax_in, float inv_h_max) { return expanded_slot_scales12(amax_in, inv_h_max); }
  static __device__ __forceinline__ float slot_scale(const Scales& sc, int j) { return sc.s[j]; }
  // x: the 12 jets of the lane's pair
  static __device__ __forceinline__ void state(const float* knots, int nk, const f32x4 (&x)[3], const Scales& sc, State& st,
                                               unsigned (&idx)[1]) {
    idx[0] = sp_pair<C1, ACT, GEN>(knots, nk, x[0][0], st.pr);
    st.rc = row_coefs12(x, sc);
  }
  // slot J of the pair's A operands (J is a constant after unrolling): word g = (live column of group g, second element)
  static __device__ __forceinline__ void cut(const State& st, SpA12& A, const int J) {
    const int q = J >> 2, r = J & 3;
    auto val = [&](float b0, float b1, float b2) {
      if (J == 0) return st.rc.s0 * b0;
      float v = st.rc.be[q][r] * b1;
      if (J == 7) v += st.rc.ga[1][3] * b2;
      if (q == 2) v += st.rc.ga[2][r] * b2;
      return v;
    };
    const float cb = val(st.pr.f0, st.pr.f1, st.pr.f2);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float cg = val(st.pr.b[g], st.pr.d1[g], st.pr.d2[g]);
      unsigned h, l = A.l[J][g];               // (groups 0, 2, 3: high half zero since the kernel's start)
      if (g == 1) split_pair16(cg, cb, h, l); else split_single16(cg, h, l);
      A.h[J][g] = h;
      A.l[J][g] = l;
    }
  }
};

// row tiles j0, j0 + 1 against output tiles n0, n0 + 1: 12 matrix instructions on 4 accumulators
template <int NIDX>
__device__ __forceinline__ void sp_multiply12(f32x4 (&acc)[12][4], const SpA12& A, const unsigned (&idx)[NIDX], const SpW& W0,
                                              const SpW& W1, const int j0, const int n0) {
  const f16x16 Bh[2] = {sp_b16(W0.h[0], W0.h[1]), sp_b16(W1.h[0], W1.h[1])};
  const f16x16 Bl[2] = {sp_b16(W0.l[0], W0.l[1]), sp_b16(W1.l[0], W1.l[1])};
  const int ix = (int)idx[j0 * NIDX / 12];                 // (j0 is even: both row tiles have the same positions)
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
      acc[j0 + jj][n0 + nn] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(__builtin_bit_cast(f16x8, A.l[j0 + jj]), Bh[nn],
                                                                       acc[j0 + jj][n0 + nn], ix, 0, 0);
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
      acc[j0 + jj][n0 + nn] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(__builtin_bit_cast(f16x8, A.h[j0 + jj]), Bl[nn],
                                                                       acc[j0 + jj][n0 + nn], ix, 0, 0);
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
      acc[j0 + jj][n0 + nn] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(__builtin_bit_cast(f16x8, A.h[j0 + jj]), Bh[nn],
                                                                       acc[j0 + jj][n0 + nn], ix, 0, 0);
}

// One stage.  On entry: A = row tiles 0 .. 9 of this stage's pairs, st = the state they were cut from, idx = their
// positions; xuse = the input jets of the next stage's pairs.  On exit the same for the next stage, and xuse holds
// the jets three stages on (the caller alternates two such buffers).
template <class M>
__device__ __forceinline__ void sp12_stage(f32x4 (&acc)[12][4], SpW (&Wr)[4], SpA12& A, unsigned (&idx)[M::NIDX],
                                           typename M::State& st, f32x4 (&xuse)[3], const u32x4* __restrict__ wnext,
                                           const float* knots_next, int nk, const f32x4* (&xload)[3],
                                           const typename M::Scales& sc, int lane) {
  sp_multiply12<M::NIDX>(acc, A, idx, Wr[0], Wr[1], 0, 0);
  M::cut(st, A, 10);
  M::cut(st, A, 11);
  interleave<12, 3>();
  PDEKAN_REGION_FENCE();
#pragma unroll
  for (int u = 1; u < 6; ++u) sp_multiply12<M::NIDX>(acc, A, idx, Wr[0], Wr[1], 2 * u, 0);
  unsigned idx_next[M::NIDX];
  M::state(knots_next, nk, xuse, sc, st, idx_next);
  interleave<60, M::NIDX == 1 ? 3 : 6>();
  PDEKAN_REGION_FENCE();
  sp_load_w(Wr[0], wnext, 0, lane);
  sp_load_w(Wr[1], wnext, 1, lane);
  sp_multiply12<M::NIDX>(acc, A, idx, Wr[2], Wr[3], 0, 2);
#pragma unroll
  for (int q = 0; q < 3; ++q) xuse[q] = *xload[q];
  interleave<12, 2>();
  PDEKAN_REGION_FENCE();
  // (the row tiles cut below belong to the next stage: idx keeps this stage's positions until the stage is issued)
#pragma unroll
  for (int u = 7; u < 12; ++u) {
    sp_multiply12<M::NIDX>(acc, A, idx, Wr[2], Wr[3], 2 * (u - 6), 2);
    M::cut(st, A, 2 * (u - 7));
    M::cut(st, A, 2 * (u - 7) + 1);
    interleave<12, 3>();
    PDEKAN_REGION_FENCE();
  }
  sp_load_w(Wr[2], wnext, 2, lane);
  sp_load_w(Wr[3], wnext, 3, lane);
#pragma unroll
  for (int t = 0; t < M::NIDX; ++t) idx[t] = idx_next[t];
}

template <class M, int C1, bool GEN>
__device__ __forceinline__ void
fwd_sp12_body(const f32x4* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
              const u32x4* __restrict__ wsp, f32x4* __restrict__ y, int64_t N, int I, int O,
              const unsigned* __restrict__ amax_in, const unsigned* __restrict__ wmax, unsigned* __restrict__ amax_out) {
  constexpr int nk = C1 + 3, KP = layer_knot_pad(nk, GEN);
  constexpr int PTS = M::POINTS, V = M::VECS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  float* sK = reinterpret_cast<float*>(smem);                               // [I][KP], shared by the block
  if constexpr (GEN) {                       // [g_0 x3][the nk knots][g_last x4][1/h]
    for (int idx = tid; idx < I * KP; idx += 256) {
      const int f = idx / KP, k = idx - f * KP - 3;
      sK[idx] = (k == nk + 4) ? ginfo[f] : grid[(size_t)f * nk + (k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k))];
    }
  } else {
    for (int idx = tid; idx < I * (nk + 1); idx += 256) {
      const int f = idx / (nk + 1), k = idx - f * (nk + 1);
      sK[f * KP + k] = (k < nk) ? grid[(size_t)f * nk + k] : ginfo[f];
    }
  }
  const typename M::Scales sc = M::scales(amax_in, max_inv_h(ginfo, I, lane));
  const float sw = pow2_scale(__uint_as_float(wmax[0]));
  __syncthreads();

  const int oc = lane & 15, tb = lane >> 4;                // expansion: point oc of a 16-point tile, feature tb of the stage
  const int o0 = blockIdx.y * 64;
  const int nst = I / 4;                                   // (at least 4, and even)
  const u32x4* wbase = wsp + (size_t)blockIdx.y * nst * kSpStageVecs;
  const float* kbase = sK + tb * KP + (GEN ? 3 : 0);
  // the lane's 3 vectors of a stage: the 12 jets of its pair, or the 4 jets of its feature at three points
  struct Rows { const f32x4* r[3]; };
  auto rows_of = [&](int64_t tile) {
    Rows R;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      int64_t prow = (tile * 4 + wave) * PTS + (V == 3 ? 0 : 16 * q) + oc;
      if (prow > N - 1) prow = N - 1;
      R.r[q] = a + (prow * I + tb) * V + (V == 3 ? q : 0);   // + 4 V per stage
    }
    return R;
  };
  int64_t tile = blockIdx.x;
  if ((tile * 4 + wave) * PTS >= N) return;                // wave-uniform; no barrier below
  Rows xrow = rows_of(tile);

  f32x4 acc[12][4];
#pragma unroll
  for (int j = 0; j < 12; ++j)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 vmax = {0.f, 0.f, 0.f, 0.f};
  SpW Wr[4];
  SpA12 A;
#pragma unroll
  for (int j = 0; j < 12; ++j) A.l[j] = (u32x4){0u, 0u, 0u, 0u};    // (M::cut keeps the high halves)
  typename M::State st;
  unsigned idx[M::NIDX];
  f32x4 xa[3], xb[3];
#pragma unroll
  for (int n = 0; n < 4; ++n) sp_load_w(Wr[n], wbase, n, lane);
  {
    f32x4 x0[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { x0[q] = xrow.r[q][0]; xa[q] = xrow.r[q][4 * V]; xb[q] = xrow.r[q][8 * V]; }
    M::state(kbase, nk, x0, sc, st, idx);
#pragma unroll
    for (int j = 0; j < 10; ++j) M::cut(st, A, j);
  }
  for (;;) {
    const int64_t P0 = (tile * 4 + wave) * PTS;
    const int64_t tnext = tile + gridDim.x;
    const bool more = (tnext * 4 + wave) * PTS < N;
    const Rows xrow_next = more ? rows_of(tnext) : xrow;   // (past the last tile: any valid data, never multiplied)
    for (int s = 0; s < nst; s += 2) {                     // (an even number of stages)
      const int s1 = s + 1, s2 = (s + 2 < nst) ? s + 2 : s + 2 - nst;
      const f32x4 *x3[3], *x4[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        x3[q] = (s + 3 < nst) ? xrow.r[q] + 4 * V * (s + 3) : xrow_next.r[q] + 4 * V * (s + 3 - nst);
        x4[q] = (s + 4 < nst) ? xrow.r[q] + 4 * V * (s + 4) : xrow_next.r[q] + 4 * V * (s + 4 - nst);
      }
      sp12_stage<M>(acc, Wr, A, idx, st, xa, wbase + (size_t)s1 * kSpStageVecs, kbase + 4 * s1 * KP, nk, x3, sc, lane);
      sp12_stage<M>(acc, Wr, A, idx, st, xb, wbase + (size_t)s2 * kSpStageVecs, kbase + 4 * s2 * KP, nk, x4, sc, lane);
    }
    // D: this lane holds rows 4 tb + i of every row tile for output 16n + oc
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      f32x4 inv;
#pragma unroll
      for (int j = 0; j < 4; ++j) inv[j] = rcp_pow2(M::slot_scale(sc, 4 * g + j) * sw);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t p = P0 + (V == 3 ? 0 : 16 * g) + 4 * tb + i;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const f32x4 v = (f32x4){acc[4 * g][n][i], acc[4 * g + 1][n][i], acc[4 * g + 2][n][i], acc[4 * g + 3][n][i]} * inv;
          if (p < N) {
            y[(p * O + o0 + 16 * n + oc) * V + (V == 3 ? g : 0)] = v;
            if constexpr (V == 1) {
#pragma unroll
              for (int j = 0; j < 4; ++j) vmax[j] = fmaxf(vmax[j], fabsf(v[j]));
            }
          }
        }
      }
    }
    if (!more) break;
#pragma unroll
    for (int j = 0; j < 12; ++j)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    tile = tnext;
    xrow = xrow_next;
  }
  if constexpr (V == 1) {
    if (amax_out) publish_jet_max(vmax, amax_out, lane);
  }
}

template <int C1, int ACT>
__global__ void __launch_bounds__(256, 1)
fwd12_sp_f16_kernel(const f32x4* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
                    const u32x4* __restrict__ wsp, f32x4* __restrict__ y, int64_t N, int I, int O,
                    const unsigned* __restrict__ amax_in, const unsigned* __restrict__ wmax) {
  const int cls = layer_knot_class(ginfo, I, threadIdx.x & 63);
  if (cls == 1) fwd_sp12_body<SpRows12<C1, ACT, false>, C1, false>(a, grid, ginfo, wsp, y, N, I, O, amax_in, wmax, nullptr);
  else if (cls == 2) fwd_sp12_body<SpRows12<C1, ACT, true>, C1, true>(a, grid, ginfo, wsp, y, N, I, O, amax_in, wmax, nullptr);
}

// ---- input-jet gradients, 12 jets: wave = 8 points x 3 groups = 6 row tiles of gY (A operands, in registers) against
// the W planes of 64 outputs at a time, as in bwd_input_f16_kernel; a stage = 4 features = NB column tiles x 2
// k-steps = 2*NB regions of 18 matrix instructions.  The stage's gE tile [8 points][4*CP columns][12 slots] is
// parked in LDS; lane (p = lane & 7, f = (lane >> 3) & 3) pulls the 4 + 1 live columns of its pair and applies the
// chain rule during the matrix instructions of the next stage (lanes 32..63 repeat lanes 0..31).
template <int C1, int ACT, bool GEN>
struct Bwi12 {
  static constexpr int ROW = BwiTiles<C1>::RING * 12 + 4;  // words per point in the parked gE tile (+4: bank spread)
  static constexpr int GROW = 64 * 12 + 4;                 // words per point of the staged raw gY rows
  static constexpr int TILE = 8 * (ROW > GROW ? ROW : GROW);
  static constexpr int KP = GEN ? kKnotPadN : kKnotPadU;
  static constexpr int nk = C1 + 3;
};

struct PairBack12 { Basis bs; Act av; f32x4 x[3]; };

// gradient with respect to the 12 input jets of one pair from the gE jets q of its 4 + 1 live columns, with
//   E_0 = B,  E_{1+k} = B' a_{1+k},  E_{7+k} = B'' a_{1+k}^2 + B' a_{7+k}:
//   d/da_0     = sum_m q_0 B' + (sum_k q_{1+k} a_{1+k} + q_{7+k} a_{7+k}) B'' + (sum_k q_{7+k} a_{1+k}^2) B'''
//   d/da_{1+k} = sum_m q_{1+k} B' + 2 a_{1+k} q_{7+k} B''          d/da_{7+k} = sum_m q_{7+k} B'
// inv[j] takes the scales of slot j out of q.
__device__ __forceinline__ void chain_rule12(const PairBack12& pb, const f32x4 (&dcol)[5][3], const float (&inv)[12],
                                             f32x4 (&out)[3]) {
  // sums over the columns, whole groups of four slots at a time (packed FMAs); U3 only for the second-derivative slots
  f32x4 U1[3], U2[3], U3b = {0.f, 0.f, 0.f, 0.f};
  float U3a = 0.f;
#pragma unroll
  for (int g = 0; g < 3; ++g) { U1[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; U2[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const float e1 = m < 4 ? pb.bs.d1[m & 3] : pb.av.f1;
    const float e2 = m < 4 ? pb.bs.d2[m & 3] : pb.av.f2;
    const float e3 = m < 4 ? pb.bs.d3[m & 3] : pb.av.f3;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      U1[g] += dcol[m][g] * e1;
      U2[g] += dcol[m][g] * e2;
    }
    U3a += dcol[m][1][3] * e3;
    U3b += dcol[m][2] * e3;
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const f32x4 iv = {inv[4 * g], inv[4 * g + 1], inv[4 * g + 2], inv[4 * g + 3]};
    U1[g] *= iv;
    U2[g] *= iv;
    if (g == 2) U3b *= iv;
  }
  U3a *= inv[7];
  const float U3[5] = {U3a, U3b[0], U3b[1], U3b[2], U3b[3]};
  float a[12], u2[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) { a[j] = pb.x[j >> 2][j & 3]; u2[j] = U2[j >> 2][j & 3]; }
  float o0 = U1[0][0];
#pragma unroll
  for (int k = 0; k < 6; ++k) o0 += a[1 + k] * u2[1 + k];
#pragma unroll
  for (int k = 0; k < 5; ++k) o0 += a[7 + k] * u2[7 + k] + a[1 + k] * a[1 + k] * U3[k];
#pragma unroll
  for (int g = 0; g < 3; ++g) out[g] = U1[g];
  out[0][0] = o0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int j = 1 + k;
    out[j >> 2][j & 3] += 2.0f * a[1 + k] * u2[7 + k];
  }
}

// N must be a multiple of 8.  cot_in: per-slot maxima of gy (12 words).
template <int C1, int ACT, bool GEN>
__global__ void __launch_bounds__(256, 1)
bwd_input12_f16_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ gy, const float* __restrict__ grid,
                       const float* __restrict__ ginfo, const u32x4* __restrict__ wpl, f32x4* __restrict__ ga, int64_t N,
                       int I, int O, const unsigned* __restrict__ cot_in, const unsigned* __restrict__ wmax) {
  using F = Bwi12<C1, ACT, GEN>;
  using T = BwiTiles<C1>;
  constexpr int KP = F::KP, nk = F::nk, ROW = F::ROW, GROW = F::GROW;
  constexpr int C = C1 - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  {
    bool near = true, equal = true;
    for (int i = lane; i < ((I + 63) & ~63); i += 64) {
      const float gi = ginfo[i < I ? i : 0];
      near = near && (__builtin_amdgcn_ballot_w64(gi != 0.0f) == ~0ull);
      equal = equal && (__builtin_amdgcn_ballot_w64(gi > 0.0f) == ~0ull);
    }
    if (!near || equal == GEN) return;
  }
  float* sK = reinterpret_cast<float*>(smem);
  float* sD = sK + I * KP + wave * F::TILE;
  if constexpr (GEN) {
    for (int idx = tid; idx < I * KP; idx += 256) {
      const int f = idx / KP, k = idx - f * KP - 3;
      sK[idx] = (k == nk + 4) ? ginfo[f] : grid[(size_t)f * nk + (k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k))];
    }
  } else {
    for (int idx = tid; idx < I * (nk + 1); idx += 256) {
      const int f = idx / (nk + 1), k = idx - f * (nk + 1);
      sK[f * KP + k] = (k < nk) ? grid[(size_t)f * nk + k] : ginfo[f];
    }
  }
  float inv[12];
  const int r = lane & 15, kb = lane >> 4, p_local = lane & 7, f_local = (lane >> 3) & 3;
  float my_t[3];                              // this lane's A-operand rows: slot 4g + (r & 3) of point (r >> 2) of the row tile
  {
    const float sw = pow2_scale(__uint_as_float(wmax[0]));
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const float t = pow2_scale(__uint_as_float(cot_in[j]));
      inv[j] = rcp_pow2(t * sw);
      if ((j & 3) == (r & 3)) my_t[j >> 2] = t;
    }
  }
  __syncthreads();

  const int64_t P0 = ((int64_t)blockIdx.x * 4 + wave) * 8;
  if (P0 >= N) return;                        // wave-uniform; no barrier below
  const int nst = I / 4;
  const int64_t prow = P0 + p_local;
  const char* xrow = reinterpret_cast<const char*>(a + (prow * I + f_local) * 3);
  f32x4* orow = ga + (prow * I + f_local) * 3;
  constexpr int kStageBytes = 4 * 48, kStageVec = 12;       // 4 features x 12 floats
  const float* kbase = sK + f_local * KP + (GEN ? 3 : 0);
  const int npass = O / 64;

  for (int ob = 0; ob < npass; ++ob) {
    // ---- the wave's gY rows of this output block -> A operands in registers ----
    u32x4 GAh[6][2], GAl[6][2];
    {
      f32x4 graw[8][3];
#pragma unroll
      for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int g = 0; g < 3; ++g) graw[v][g] = gy[((P0 + v) * O + ob * 64 + lane) * 3 + g];
#pragma unroll
      for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int g = 0; g < 3; ++g) *reinterpret_cast<f32x4*>(sD + v * GROW + lane * 12 + 4 * g) = graw[v][g];
#pragma unroll
      for (int mt = 0; mt < 6; ++mt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int g = mt >> 1, pq = mt & 1;
          const float* src = sD + (4 * pq + (r >> 2)) * GROW + (32 * ks + 8 * kb) * 12 + 4 * g + (r & 3);
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = src[e * 12] * my_t[g];
          unsigned h[4], l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split_pair16(v[2 * e], v[2 * e + 1], h[e], l[e]);
          GAh[mt][ks] = (u32x4){h[0], h[1], h[2], h[3]};
          GAl[mt][ks] = (u32x4){l[0], l[1], l[2], l[3]};
        }
    }
    const u32x4* wbase = wpl + (size_t)ob * (nst / 2) * T::kPairVecs;
    WTile16 Wt[4];
    auto fetch_w = [&](WTile16& w, const u32x4* src) { w.h = src[lane]; w.l = src[64 + lane]; };
    fetch_w(Wt[0], wbase);
    fetch_w(Wt[1], wbase + 128);
    fetch_w(Wt[2], wbase + 256);

    f32x4 acc[6][T::TA];
    PairBack12 prev, cur;
    f32x4 dcol[5][3];
    f32x4 old[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (ob > 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g) old[g] = orow[g];
    }
    prev.av = Act{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) prev.bs.d1[m] = prev.bs.d2[m] = prev.bs.d3[m] = 0.0f;
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
      for (int g = 0; g < 3; ++g) dcol[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 xnext[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      prev.x[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      cur.x[g] = *reinterpret_cast<const f32x4*>(xrow + 16 * g);
      xnext[g] = *reinterpret_cast<const f32x4*>(xrow + (nst > 1 ? kStageBytes : 0) + 16 * g);
    }

    // one stage: NBs column tiles, B-tile ring starting at slot PH, ODD: the second stage of its pair
    auto stage = [&](auto nb_c, auto ph_c, auto odd_c, int s, const u32x4* wsrc, const u32x4* wnxt)
                     __attribute__((always_inline)) {
      constexpr int NBs = decltype(nb_c)::value, PH = decltype(ph_c)::value, NRs = 2 * NBs;
      constexpr bool ODD = decltype(odd_c)::value;
      f32x4* odst = orow + (s > 0 ? s - 1 : 0) * kStageVec;
#pragma unroll
      for (int i = 0; i < NRs; ++i) {
        const int ks = i / NBs, n = i % NBs;
        {
          const int t = i + 3;                // three regions ahead, into the slot the last region left
          fetch_w(Wt[(t + PH) & 3], (t < NRs) ? wsrc + t * 128 : wnxt + (t - NRs) * 128);
        }
        if (i == 0) {                         // chain rule of the stage before, on the columns read after its park
          f32x4 out[3];
          chain_rule12(prev, dcol, inv, out);
#pragma unroll
          for (int g = 0; g < 3; ++g) odst[g] = out[g] + old[g];
        }
        if (i == 1) {                         // this stage's pair: basis, activation (kept for the next stage)
          const float* kn = kbase + 4 * s * KP;
          const FeatGrid fg = {kn[0], kn[nk - 1], fabsf(kn[GEN ? nk + 4 : nk])};
          eval_pair_d3<ACT, GEN>(kn, nk, fg, cur.x[0][0], cur.bs, cur.av);
          if (ob > 0) {
#pragma unroll
            for (int g = 0; g < 3; ++g) old[g] = orow[s * kStageVec + g];
          }
        }
        if (i > NBs) {                        // the column tile the region before completed: park it (see below)
#pragma unroll
          for (int m = 0; m < 6; ++m)
            *reinterpret_cast<f32x4*>(sD + (4 * (m & 1) + kb) * ROW + (16 * (n - 1) + r) * 12 + 4 * (m >> 1)) = acc[m][n - 1];
        }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 6; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, GAl[m][ks]),
                                                             __builtin_bit_cast(f16x8, Wt[(i + PH) & 3].h),
                                                             ks == 0 ? zero4 : acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 6; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, GAh[m][ks]),
                                                             __builtin_bit_cast(f16x8, Wt[(i + PH) & 3].l), acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 6; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, GAh[m][ks]),
                                                             __builtin_bit_cast(f16x8, Wt[(i + PH) & 3].h), acc[m][n], 0, 0, 0);
        if (i == 0) interleave<18, 14>(); else if (i == 1) interleave<18, 9>(); else interleave<18, 2>();
        PDEKAN_REGION_FENCE();
      }
      // park the stage's gE tile: lane (c = r, q = kb) holds slots 4g .. 4g+3 of point 4pq + q, column slot 16n + c
      // (the tiles before the last one went out during the regions above)
#pragma unroll
      for (int m = 0; m < 6; ++m)
        *reinterpret_cast<f32x4*>(sD + (4 * (m & 1) + kb) * ROW + (16 * (NBs - 1) + r) * 12 + 4 * (m >> 1)) = acc[m][NBs - 1];
      const float* drow = sD + p_local * ROW;
      const int g0 = (ODD ? T::OFFB : 0) + f_local * C1;
      auto slot_of = [&](int c) { return T::WRAP ? (g0 + c) % T::RING : f_local * C1 + c; };
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int c = cur.bs.c0 + m;
        const bool ok = (c >= 0) & (c < C);
        const int slot = slot_of(ok ? c : C);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(drow + slot * 12 + 4 * g);
          dcol[m][g] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int g = 0; g < 3; ++g) dcol[4][g] = *reinterpret_cast<const f32x4*>(drow + slot_of(C) * 12 + 4 * g);
      prev = cur;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        cur.x[g] = xnext[g];
        xnext[g] = *reinterpret_cast<const f32x4*>(xrow + (size_t)((s + 2 < nst) ? s + 2 : 0) * kStageBytes + 16 * g);
      }
    };
    using std::integral_constant;
    if constexpr (T::TA == T::TB && T::PH1 == 0 && T::PH2 == 0 && T::OFFB % T::RING == 0) {
      // every stage is whole tiles of its own and leaves the ring where it found it: one stage body
      for (int s = 0; s < nst; ++s) {
        const u32x4* pA = wbase + (size_t)s * (T::kPairVecs / 2);
        stage(integral_constant<int, T::TA>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}, s, pA,
              (s + 1 < nst) ? pA + T::kPairVecs / 2 : wbase);
      }
    } else
    for (int s = 0; s < nst; s += 4) {          // two pairs of stages (nst is even)
      const bool more = s + 2 < nst;
      const u32x4* pA = wbase + (size_t)(s >> 1) * T::kPairVecs;
      const u32x4* pB = pA + 2 * T::TA * 128;
      const u32x4* pC = more ? pA + T::kPairVecs : wbase;            // past the end: any valid planes, never multiplied
      stage(integral_constant<int, T::TA>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}, s, pA, pB);
      stage(integral_constant<int, T::TB>{}, integral_constant<int, T::PH1>{}, integral_constant<bool, true>{}, s + 1, pB, pC);
      if (more) {
        const u32x4* pD = pC + 2 * T::TA * 128;
        const u32x4* pE = (s + 4 < nst) ? pC + T::kPairVecs : wbase;
        stage(integral_constant<int, T::TA>{}, integral_constant<int, T::PH2>{}, integral_constant<bool, false>{}, s + 2, pC, pD);
        stage(integral_constant<int, T::TB>{}, integral_constant<int, T::PH3>{}, integral_constant<bool, true>{}, s + 3, pD, pE);
      }
    }
    {                                           // chain rule of the last stage
      f32x4 out[3];
      chain_rule12(prev, dcol, inv, out);
#pragma unroll
      for (int g = 0; g < 3; ++g) orow[(nst - 1) * kStageVec + g] = out[g] + old[g];
    }
  }
}

// ---- weight gradients, 12 jets: the kernel above over virtual points.  A block of 8 points is three steps, one per
// group; step (b, g) multiplies the E tile [columns][8 points x 4 rows of group g] with the gY rows of the same 8
// points and group (K = 32).  The basis of a pair is evaluated once per block (with group 0) and kept for the other
// two groups, whose columns differ only in the row coefficients.  Scales: gY's slot j by t_j, E's by c / t_j.
// Runs when every knot row of the layer is (nearly) equally spaced; a block takes the variant that matches its own
// features' rows.  chunk_pts and N are multiples of 16.
constexpr int kF16FlushPoints12 = 160;       // 60 chained steps per accumulator between flushes (512 / 8 = 64 above)

struct PairKeep12 { Basis bs; Act av; };

// columns of group GRP of one pair, scaled
template <int GRP>
__device__ __forceinline__ void expand_group12(const PairKeep12& pk, const f32x4 (&x)[3], const SlotScales12& sc, PairCols& out) {
  const RowCoefs12 rc = row_coefs12(x, sc);
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const float b0 = m < 4 ? pk.bs.b[m & 3] : pk.av.f0, b1 = m < 4 ? pk.bs.d1[m & 3] : pk.av.f1;
    const float b2 = m < 4 ? pk.bs.d2[m & 3] : pk.av.f2;
    if (GRP == 0) { out.col[m] = rc.be[0] * b1; out.col[m][0] = rc.s0 * b0; }
    if (GRP == 1) { out.col[m] = rc.be[1] * b1; out.col[m][3] += rc.ga[1][3] * b2; }
    if (GRP == 2) out.col[m] = rc.be[2] * b1 + rc.ga[2] * b2;
  }
  out.c0 = pk.bs.c0;
}

template <int ACT, bool GEN>
__device__ __forceinline__ void eval_keep12(const float* knots, int nk, const FeatGrid& fg, float x, PairKeep12& pk) {
  if constexpr (GEN) eval_basis_near_uniform(knots, nk, fg.g0, fg.glast, fg.inv_h, x, pk.bs);
  else eval_basis_uniform_by_quotient(knots, nk, fg.g0, fg.glast, fg.inv_h, x, pk.bs);
  pk.av = eval_act(ACT, x);
}

// One step.  GN = the group of the step being prepared (s+1).  graw holds the raw gY rows of step s+1 on entry,
// xnext the jets of the pairs of the block after the one in xcur.  Region plan as in wg16_step; with GN == 0 the
// pairs of the next block move in (xcur <- xnext, basis evaluated) and the block after it is fetched from xrow2.

template <int C1, int FE, int ACT, bool GEN, int GN>
__device__ __forceinline__ void wg12_step(f32x4 (&acc)[FE * C1 / 16][4], const GyPlanes16& G, GyPlanes16& Gn, TileOps16& cur,
                                          f32x4 (&graw)[4][2], f32x4 (&xcur)[3], f32x4 (&xnext)[3], PairKeep12& pk,
                                          unsigned* sE, const float* knots, const FeatGrid& fg, const SlotScales12& se,
                                          const f32x4& tg, int& prev_c0, const char* __restrict__ xrow2,
                                          const char* __restrict__ grow1, const char* __restrict__ grow2, unsigned xoff,
                                          unsigned goff, int lane, int O) {
  constexpr int COLS = FE * C1, NT = COLS / 16, nk = C1 + 3;
  static_assert(NT == 6 || NT == 7, "the region plan below is written for 6 or 7 column tiles");
  const int r = lane & 15, q = lane >> 4, p_local = lane & 7, f_local = (lane >> 3) % FE;
  const unsigned* esrc = sE + r * kWgRw + q * 4;
  PairCols pj;
  Planes16 pp[5];
  TileOps16 last;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    TileOps16 nxt;
    if (t + 1 < NT - 1) nxt = read_wg_tile<COLS>(esrc, t + 1);
    if (t == NT - 3) last = read_wg_tile<COLS>(esrc, NT - 1);
    if (t == NT - 2) nxt = last;
    if (t == 0) {
      if (GN == 0) {
#pragma unroll
        for (int g = 0; g < 3; ++g) xcur[g] = xnext[g];
        eval_keep12<ACT, GEN>(knots, nk, fg, xcur[0][0], pk);
      }
      expand_group12<GN>(pk, xcur, se, pj);
#pragma unroll
      for (int e = 0; e < 2; ++e) graw[3][e] = ld16v(grow1, goff + (unsigned)(e * O + 3 * 16) * 48u);
    }
    if (t == 1) {
      if (GN == 0) {
#pragma unroll
        for (int g = 0; g < 3; ++g) xnext[g] = ld16v(xrow2, xoff + 16u * g);
      }
      pp[0] = split_jet16(pj.col[0]); pp[1] = split_jet16(pj.col[1]); pp[2] = split_jet16(pj.col[2]);
    }
    if (t == 2) { pp[3] = split_jet16(pj.col[3]); pp[4] = split_jet16(pj.col[4]); }
    if (t >= NT - 4) split_gy_tile16(graw[t - (NT - 4)], tg, Gn, t - (NT - 4));
    if (t >= NT - 3) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
        graw[t - (NT - 3)][e] = ld16v(grow2, goff + (unsigned)(e * O + (t - (NT - 3)) * 16) * 48u);
    }
    if (t == NT - 3) {
      unsigned* base = sE + (f_local * C1) * kWgRw + p_local * 2;
      park_wg16<C1, COLS>(base, sE + (f_local * C1) * kWgRw + 16, pp, pj.c0, prev_c0);
    }
    if (t == NT - 1) nxt = read_wg_tile<COLS>(esrc, 0);
    mfma_wg_tile(acc[t], cur, G);
    cur = nxt;
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int C1, int FE, int ACT, bool GEN>
__global__ void __launch_bounds__(256, 1)
bwd_weight12_f16_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ gy, const float* __restrict__ grid,
                        const float* __restrict__ ginfo, float* __restrict__ part, int64_t N, int I, int O,
                        int64_t chunk_pts, int nx, int ny, int nz, const unsigned* __restrict__ act_in,
                        const unsigned* __restrict__ cot_in) {
  constexpr int COLS = FE * C1, NT = COLS / 16, nk = C1 + 3;
  static_assert(COLS % 16 == 0 && (FE == 8 || FE == 4), "16-wide column tiles; 8 points x FE features per step");
  constexpr int KP = GEN ? kKnotPadN : kKnotPadU;
  constexpr int kWaveWords = 2 * COLS * kWgRw + FE * KP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bx, by, bz;
  {
    const int b = blockIdx.x;
    if (nz % 8 == 0) {
      const int rr = b >> 3;
      bx = rr % nx; by = (rr / nx) % ny; bz = (rr / (nx * ny)) * 8 + (b & 7);
    } else {
      bx = b % nx; by = (b / nx) % ny; bz = b / (nx * ny);
    }
  }
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if (!all_rows_near_uniform(ginfo, I)) return;         // the split passes of the 4-jet kernels do the layer
  {
    const float gi = ginfo[bx * 4 * FE + lane % (4 * FE)];
    const bool all_equal = __builtin_amdgcn_ballot_w64(gi > 0.0f) == ~0ull;
    if (all_equal == GEN) return;
  }
  unsigned* sE = reinterpret_cast<unsigned*>(smem) + wave * kWaveWords;
  float* sK = reinterpret_cast<float*>(sE + 2 * COLS * kWgRw);
  const int r = lane & 15, q = lane >> 4, p_local = lane & 7, f_local = (lane >> 3) % FE;
  const int i0 = (bx * 4 + wave) * FE, o0 = by * 64;
  if constexpr (GEN) {
    for (int idx = lane; idx < FE * KP; idx += 64) {
      const int f = idx / KP, k = idx - f * KP - 3;
      sK[idx] = grid[(size_t)(i0 + f) * nk + (k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k))];
    }
  } else {
    for (int idx = lane; idx < FE * (nk + 1); idx += 64) {
      const int f = idx / (nk + 1), k = idx - f * (nk + 1);
      sK[f * KP + k] = (k < nk) ? grid[(size_t)(i0 + f) * nk + k] : ginfo[i0 + f];
    }
  }
  for (int idx = lane; idx < 2 * COLS * kWgRw; idx += 64) sE[idx] = 0u;
  const float* knots = sK + f_local * KP + (GEN ? 3 : 0);
  const FeatGrid fg = {grid[(size_t)(i0 + f_local) * nk], grid[(size_t)(i0 + f_local) * nk + nk - 1],
                       fabsf(ginfo[i0 + f_local])};
  SlotScales12 se;
  f32x4 tg[3];
  float inv_c;
  {
    const SlotScales12 eb = expanded_slot_scales12(act_in, max_inv_h(ginfo, I, lane));
    float c = 3.0e38f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      tg[j >> 2][j & 3] = pow2_scale(__uint_as_float(cot_in[j]));
      c = fminf(c, eb.s[j] * tg[j >> 2][j & 3]);
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) se.s[j] = c * rcp_pow2(tg[j >> 2][j & 3]);
    inv_c = rcp_pow2(c);
  }

  f32x4 acc[NT][4], tot[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) { acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f}; tot[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  const int64_t pbeg = (int64_t)bz * chunk_pts;
  const int64_t pend = (pbeg + chunk_pts < N) ? pbeg + chunk_pts : N;
  const char* xb = reinterpret_cast<const char*>(a);
  const char* gb = reinterpret_cast<const char*>(gy);
  const unsigned xoff = (unsigned)(p_local * I + i0 + f_local) * 48u;
  const unsigned goff = (unsigned)(2 * q * O + o0 + r) * 48u;
  const size_t xstride = (size_t)I * 48, gstride = (size_t)O * 48;       // bytes per point

  if (pbeg < pend) {
    f32x4 graw[4][2], xcur[3], xnext[3];
    GyPlanes16 GA, GB;
    PairKeep12 pk;
    int prev_c0 = -100;
    {
      f32x4 g0[4][2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          g0[m][e] = ld16v(gb + pbeg * gstride, goff + (unsigned)(e * O + m * 16) * 48u);
          graw[m][e] = ld16v(gb + pbeg * gstride + 16, goff + (unsigned)(e * O + m * 16) * 48u);
        }
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        xcur[g] = ld16v(xb + pbeg * xstride, xoff + 16u * g);
        xnext[g] = ld16v(xb + (pbeg + 8) * xstride, xoff + 16u * g);
      }
      PairCols pj;
      Planes16 pp[5];
      eval_keep12<ACT, GEN>(knots, nk, fg, xcur[0][0], pk);
      expand_group12<0>(pk, xcur, se, pj);
#pragma unroll
      for (int m = 0; m < 5; ++m) pp[m] = split_jet16(pj.col[m]);
      unsigned* base = sE + (f_local * C1) * kWgRw + p_local * 2;
      park_wg16<C1, COLS>(base, sE + (f_local * C1) * kWgRw + 16, pp, pj.c0, prev_c0);
#pragma unroll
      for (int m = 0; m < 4; ++m) split_gy_tile16(g0[m], tg[0], GA, m);
    }
    TileOps16 cur = read_wg_tile<COLS>(sE + r * kWgRw + q * 4, 0);
    for (int64_t pb = pbeg; pb < pend;) {
      const int64_t pstop = (pb + kF16FlushPoints12 < pend) ? pb + kF16FlushPoints12 : pend;
      for (; pb < pstop; pb += 16) {
        // past the end of the chunk: any valid rows, fetched (and expanded) but never used
        const int64_t p1 = pb + 8, p2 = (pb + 16 < pend) ? pb + 16 : pbeg, p3 = (pb + 24 < pend) ? pb + 24 : pbeg;
        const char* gA = gb + pb * gstride;     // gY rows of the four blocks around: this, next, the two after
        const char* gB = gb + p1 * gstride;
        const char* gC = gb + p2 * gstride;
        // step (pb, 0): prepares (pb, 1), fetches gY of (pb, 2)
        wg12_step<C1, FE, ACT, GEN, 1>(acc, GA, GB, cur, graw, xcur, xnext, pk, sE, knots, fg, se, tg[1], prev_c0, xb,
                                       gA + 16, gA + 32, xoff, goff, lane, O);
        // step (pb, 1): prepares (pb, 2), fetches gY of (p1, 0)
        wg12_step<C1, FE, ACT, GEN, 2>(acc, GB, GA, cur, graw, xcur, xnext, pk, sE, knots, fg, se, tg[2], prev_c0, xb,
                                       gA + 32, gB, xoff, goff, lane, O);
        // step (pb, 2): prepares (p1, 0) (its jets are in xnext), fetches the jets of p2 and gY of (p1, 1)
        wg12_step<C1, FE, ACT, GEN, 0>(acc, GA, GB, cur, graw, xcur, xnext, pk, sE, knots, fg, se, tg[0], prev_c0,
                                       xb + p2 * xstride, gB, gB + 16, xoff, goff, lane, O);
        wg12_step<C1, FE, ACT, GEN, 1>(acc, GB, GA, cur, graw, xcur, xnext, pk, sE, knots, fg, se, tg[1], prev_c0, xb,
                                       gB + 16, gB + 32, xoff, goff, lane, O);
        wg12_step<C1, FE, ACT, GEN, 2>(acc, GA, GB, cur, graw, xcur, xnext, pk, sE, knots, fg, se, tg[2], prev_c0, xb,
                                       gB + 32, gC, xoff, goff, lane, O);
        wg12_step<C1, FE, ACT, GEN, 0>(acc, GB, GA, cur, graw, xcur, xnext, pk, sE, knots, fg, se, tg[0], prev_c0,
                                       xb + p3 * xstride, gC, gC + 16, xoff, goff, lane, O);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          tot[t][m] += acc[t][m];
          acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float* dst = part + (((size_t)bz * O + o0 + m * 16 + r) * I + i0) * C1 + t * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(dst) = tot[t][m] * inv_c;
    }
}

// ==========================================================================================================
// host side
// ==========================================================================================================
static bool f16_disabled() {                 // A/B switch for tests and profiling
  const char* v = getenv("PDEKAN_DISABLE_F16");
  return v && v[0] == '1';
}

// layers the fp16 matrix kernels take (whether they run is settled on the device from the knot rows)
bool f16_eligible(int D, int D2, const LayerView& L) {
  if (D != 2 || D2 != 1 || f16_disabled()) return false;
  if (L.G != 10 && L.G != 20) return false;
  return L.I >= 8 && L.I % 8 == 0 && L.I <= 256 && L.O % 64 == 0 && L.O <= 256;
}

// 12-jet layers (6 directions, 5 second derivatives) the fp16 matrix kernels take
bool f16_jets12_eligible(int D, int D2, const LayerView& L) {
  if (D != 6 || D2 != 5 || f16_disabled()) return false;
  if (L.G != 10) return false;                       // one wave's E tile of a grid-20 layer does not fit beside three others
  return L.I >= 32 && L.I % 32 == 0 && L.I <= 256 && L.O % 64 == 0 && L.O <= 256;   // (32: features per weight-gradient block)
}

static int cp_of(const LayerView& L) { return (L.C1() + 7) / 8 * 8; }

// floats of workspace for the W planes of a layer (forward layout, then input-gradient layout)
size_t f16_wplane_floats(const LayerView& L) {
  const size_t fwd = (size_t)L.O * L.I * cp_of(L);        // 2 planes x CP columns x fp16 = CP floats per (o, i)
  return 2 * fwd;
}

void launch_absmax(const float* w, int64_t n, unsigned* out, hipStream_t st) {
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  absmax_kernel<<<blocks, 256, 0, st>>>(w, n, out);
}

void launch_jet_absmax(const float* t, int64_t elems, unsigned* out4, hipStream_t st) {
  int64_t blocks = (elems + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  jet_absmax_kernel<<<(unsigned)blocks, 256, 0, st>>>(reinterpret_cast<const f32x4*>(t), elems, out4);
}

void launch_jet_absmax12(const float* t, int64_t elems, unsigned* out12, hipStream_t st) {
  int64_t blocks = (elems + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  jet_absmax12_kernel<<<(unsigned)blocks, 256, 0, st>>>(reinterpret_cast<const f32x4*>(t), elems, out12);
}

// wp[O][I][C1] -> the two W plane images; wmax: bits of max |wp| (launch_absmax)
// 12-jet layers whose forward runs on 32 x 32 x 16 tiles (fwd12w_f16_kernel: its own forward image of the W planes)
bool f16_wide_forward12(const LayerView& L) {
  const char* off = getenv("PDEKAN_DISABLE_F16_WIDE");   // A/B switch for tests and profiling
  return L.G == 10 && !(off && off[0] == '1');
}

// 4-jet layers whose forward runs on the structured-sparsity instruction (fwd_sp_f16_kernel: its own forward image)
static bool f16_sparse_forward(const LayerView& L) {
  const char* off = getenv("PDEKAN_DISABLE_F16_SPARSE");  // A/B switch for tests and profiling
  return L.G == 10 && !(off && off[0] == '1');
}

// 12-jet layers whose forward runs on the structured-sparsity instruction (fwd12_sp_f16_kernel: the same image)
static bool f16_sparse_forward12(const LayerView& L) {
  const char* off = getenv("PDEKAN_DISABLE_F16_SPARSE");  // A/B switch for tests and profiling
  return L.G == 10 && (L.I / 4) * 3 <= kFwdChainMax && !(off && off[0] == '1');
}

// jets12: the planes are for the 12-jet kernels
void launch_pack_wplanes16(const LayerView& L, const float* wp, const unsigned* wmax, float* wpl, hipStream_t st, bool jets12) {
  u32x4* f = reinterpret_cast<u32x4*>(wpl);
  u32x4* b = reinterpret_cast<u32x4*>(wpl + f16_wplane_floats(L) / 2);
  const int cp = cp_of(L);
  const int nrp = (L.G == 10) ? BwiTiles<14>::NRP : BwiTiles<24>::NRP;
  const int total = (L.O / 64) * ((L.I / 4) * (cp / 8) * 4 + (L.I / 8) * nrp) * 64;
  int blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (L.G == 10 && (jets12 ? f16_sparse_forward12(L) : f16_sparse_forward(L))) pack_wplanes16_kernel<14, false, true><<<blocks, 256, 0, st>>>(wp, wmax, f, b, L.I, L.O);
  else if (L.G == 10 && jets12 && f16_wide_forward12(L)) pack_wplanes16_kernel<14, true><<<blocks, 256, 0, st>>>(wp, wmax, f, b, L.I, L.O);
  else if (L.G == 10) pack_wplanes16_kernel<14, false><<<blocks, 256, 0, st>>>(wp, wmax, f, b, L.I, L.O);
  else pack_wplanes16_kernel<24, false><<<blocks, 256, 0, st>>>(wp, wmax, f, b, L.I, L.O);
}

template <typename Kern>
static void allow_lds(Kern k, size_t bytes) {
  ensure_dynamic_lds(reinterpret_cast<const void*>(k), bytes);
}

template <int C1, int ACT>
static void fwd_f16_launch(const LayerView& L, const float* a, const float* wpl, float* y, int64_t N,
                           const unsigned* amax_in, const unsigned* wmax, unsigned* amax_out, hipStream_t st) {
  using F = FwdF16<C1, ACT, true>;             // (the nearly-equal variant's knot rows are the longer ones)
  const bool longk = (L.I / 4) * F::KS * 3 > kFwdChainMax;
  auto k = longk ? fwd_f16_kernel<C1, ACT, true> : fwd_f16_kernel<C1, ACT, false>;
  const size_t lds = ((size_t)L.I * F::KP + 4 * 2 * F::PLANE) * 4;
  allow_lds(k, lds);
  dim3 g((unsigned)((N + 63) / 64), L.O / 64);
  k<<<g, 256, lds, st>>>(reinterpret_cast<const f32x4*>(a), L.grid, L.ginfo, reinterpret_cast<const u32x4*>(wpl),
                         reinterpret_cast<f32x4*>(y), N, L.I, L.O, amax_in, wmax, amax_out);
}

template <int C1, int ACT>
static void fwd_sp_launch(const LayerView& L, const float* a, const float* wpl, float* y, int64_t N,
                          const unsigned* amax_in, const unsigned* wmax, unsigned* amax_out, hipStream_t st) {
  const bool longk = (L.I / 4) * 3 > kFwdChainMax;
  auto k = longk ? fwd_sp_f16_kernel<C1, ACT, true> : fwd_sp_f16_kernel<C1, ACT, false>;
  const size_t lds = (size_t)L.I * layer_knot_pad(C1 + 3, true) * 4;
  // blocks walk the tiles of 64 points with a stride: four rounds of resident blocks at most (256 CUs x 2 blocks)
  int64_t tiles = (N + 63) / 64;
  const int64_t cap = 2048 / (L.O / 64);
  dim3 g((unsigned)(tiles < cap ? tiles : cap), L.O / 64);
  k<<<g, 256, lds, st>>>(reinterpret_cast<const f32x4*>(a), L.grid, L.ginfo, reinterpret_cast<const u32x4*>(wpl),
                         reinterpret_cast<f32x4*>(y), N, L.I, L.O, amax_in, wmax, amax_out);
}

template <int C1>
static void fwd_f16_variants(const LayerView& L, const float* a, const float* wpl, float* y, int64_t N,
                             const unsigned* amax_in, const unsigned* wmax, unsigned* amax_out, hipStream_t st) {
  if (L.act == kActSiLU) fwd_f16_launch<C1, kActSiLU>(L, a, wpl, y, N, amax_in, wmax, amax_out, st);
  else fwd_f16_launch<C1, kActIdentity>(L, a, wpl, y, N, amax_in, wmax, amax_out, st);
}

// one launch: a block takes the variant that matches the layer's knot rows (or returns at once)
void launch_fwd_f16(const LayerView& L, const float* a, const float* wpl, float* y, int64_t N, const unsigned* amax_in,
                    const unsigned* wmax, unsigned* amax_out, hipStream_t st) {
  if (f16_sparse_forward(L)) {
    if (L.act == kActSiLU) fwd_sp_launch<14, kActSiLU>(L, a, wpl, y, N, amax_in, wmax, amax_out, st);
    else fwd_sp_launch<14, kActIdentity>(L, a, wpl, y, N, amax_in, wmax, amax_out, st);
  } else if (L.G == 10) fwd_f16_variants<14>(L, a, wpl, y, N, amax_in, wmax, amax_out, st);
  else fwd_f16_variants<24>(L, a, wpl, y, N, amax_in, wmax, amax_out, st);
}

template <int ACT, bool GEN>
static void fwd12_launch(const LayerView& L, const float* a, const float* wpl, float* y, int64_t N,
                         const unsigned* amax_in, const unsigned* wmax, hipStream_t st) {
  using F = Fwd12<14, ACT, GEN>;
  auto k = f16_wide_forward12(L) ? fwd12w_f16_kernel<14, ACT, GEN> : fwd12_f16_kernel<14, ACT, GEN>;
  const size_t lds = ((size_t)L.I * F::KP + 4 * 2 * F::PLANE) * 4;
  allow_lds(k, lds);
  dim3 g((unsigned)((N + 31) / 32), L.O / 64);
  k<<<g, 256, lds, st>>>(reinterpret_cast<const f32x4*>(a), L.grid, L.ginfo, reinterpret_cast<const u32x4*>(wpl),
                         reinterpret_cast<f32x4*>(y), N, L.I, L.O, amax_in, wmax);
}

template <int ACT>
static void fwd12_sp_launch(const LayerView& L, const float* a, const float* wpl, float* y, int64_t N,
                            const unsigned* amax_in, const unsigned* wmax, hipStream_t st) {
  auto k = fwd12_sp_f16_kernel<14, ACT>;
  const size_t lds = (size_t)L.I * layer_knot_pad(14 + 3, true) * 4;
  // blocks walk the tiles of 64 points with a stride: four rounds of resident blocks at most (256 CUs x 1 block)
  int64_t tiles = (N + 63) / 64;
  const int64_t cap = 1024 / (L.O / 64);
  dim3 g((unsigned)(tiles < cap ? tiles : cap), L.O / 64);
  k<<<g, 256, lds, st>>>(reinterpret_cast<const f32x4*>(a), L.grid, L.ginfo, reinterpret_cast<const u32x4*>(wpl),
                         reinterpret_cast<f32x4*>(y), N, L.I, L.O, amax_in, wmax);
}

// 12-jet forward (f16_jets12_eligible layers): amax_in = per-slot maxima of a (12 words).  N must be a multiple of 8.
void launch_fwd12_f16(const LayerView& L, const float* a, const float* wpl, float* y, int64_t N, const unsigned* amax_in,
                      const unsigned* wmax, hipStream_t st) {
  if (f16_sparse_forward12(L)) {
    if (L.act == kActSiLU) fwd12_sp_launch<kActSiLU>(L, a, wpl, y, N, amax_in, wmax, st);
    else fwd12_sp_launch<kActIdentity>(L, a, wpl, y, N, amax_in, wmax, st);
    return;
  }
  if (L.act == kActSiLU) {
    fwd12_launch<kActSiLU, false>(L, a, wpl, y, N, amax_in, wmax, st);
    fwd12_launch<kActSiLU, true>(L, a, wpl, y, N, amax_in, wmax, st);
  } else {
    fwd12_launch<kActIdentity, false>(L, a, wpl, y, N, amax_in, wmax, st);
    fwd12_launch<kActIdentity, true>(L, a, wpl, y, N, amax_in, wmax, st);
  }
}

template <int ACT, bool GEN>
static void bwi12_launch(const LayerView& L, const float* a, const float* gy, const float* wplb, float* ga, int64_t N,
                         const unsigned* cot_in, const unsigned* wmax, hipStream_t st) {
  using F = Bwi12<14, ACT, GEN>;
  auto k = bwd_input12_f16_kernel<14, ACT, GEN>;
  const size_t lds = ((size_t)L.I * F::KP + 4 * F::TILE) * 4;
  allow_lds(k, lds);
  k<<<(unsigned)((N + 31) / 32), 256, lds, st>>>(reinterpret_cast<const f32x4*>(a), reinterpret_cast<const f32x4*>(gy),
                                                 L.grid, L.ginfo, reinterpret_cast<const u32x4*>(wplb),
                                                 reinterpret_cast<f32x4*>(ga), N, L.I, L.O, cot_in, wmax);
}

// 12-jet input gradients (f16_jets12_eligible layers): cot_in = per-slot maxima of gy (12 words).  N must be a
// multiple of 8.
void launch_bwd_input12_f16(const LayerView& L, const float* a, const float* gy, const float* wpl, float* ga, int64_t N,
                            const unsigned* cot_in, const unsigned* wmax, hipStream_t st) {
  const float* wplb = wpl + f16_wplane_floats(L) / 2;
  if (L.act == kActSiLU) {
    bwi12_launch<kActSiLU, false>(L, a, gy, wplb, ga, N, cot_in, wmax, st);
    bwi12_launch<kActSiLU, true>(L, a, gy, wplb, ga, N, cot_in, wmax, st);
  } else {
    bwi12_launch<kActIdentity, false>(L, a, gy, wplb, ga, N, cot_in, wmax, st);
    bwi12_launch<kActIdentity, true>(L, a, gy, wplb, ga, N, cot_in, wmax, st);
  }
}

template <int C1, int ACT, bool GEN>
static void bwi_f16_launch(const LayerView& L, const float* a, const float* gy, const float* wplb, float* ga, int64_t N,
                           const unsigned* cot_in, const unsigned* wmax, unsigned* cot_out, hipStream_t st) {
  using F = BwiF16<C1, ACT, GEN>;
  auto k = (L.O > 64) ? bwd_input_f16_kernel<C1, ACT, GEN, true> : bwd_input_f16_kernel<C1, ACT, GEN, false>;
  const size_t lds = ((size_t)L.I * F::KP + 4 * F::TILE) * 4;
  allow_lds(k, lds);
  k<<<(unsigned)((N + 63) / 64), 256, lds, st>>>(reinterpret_cast<const f32x4*>(a), reinterpret_cast<const f32x4*>(gy),
                                                 L.grid, L.ginfo, reinterpret_cast<const u32x4*>(wplb),
                                                 reinterpret_cast<f32x4*>(ga), N, L.I, L.O, cot_in, wmax, cot_out);
}

template <int C1>
static void bwi_f16_variants(const LayerView& L, const float* a, const float* gy, const float* wplb, float* ga, int64_t N,
                             const unsigned* cot_in, const unsigned* wmax, unsigned* cot_out, hipStream_t st) {
  if (L.act == kActSiLU) {
    bwi_f16_launch<C1, kActSiLU, false>(L, a, gy, wplb, ga, N, cot_in, wmax, cot_out, st);
    bwi_f16_launch<C1, kActSiLU, true>(L, a, gy, wplb, ga, N, cot_in, wmax, cot_out, st);
  } else {
    bwi_f16_launch<C1, kActIdentity, false>(L, a, gy, wplb, ga, N, cot_in, wmax, cot_out, st);
    bwi_f16_launch<C1, kActIdentity, true>(L, a, gy, wplb, ga, N, cot_in, wmax, cot_out, st);
  }
}

// cot_in: per-jet maxima of gy (4 words); cot_out: where the per-jet maxima of ga are combined into, or nullptr.
// N must be a multiple of 16.
void launch_bwd_input_f16(const LayerView& L, const float* a, const float* gy, const float* wpl, float* ga, int64_t N,
                          const unsigned* cot_in, const unsigned* wmax, unsigned* cot_out, hipStream_t st) {
  const float* wplb = wpl + f16_wplane_floats(L) / 2;
  if (L.G == 10) bwi_f16_variants<14>(L, a, gy, wplb, ga, N, cot_in, wmax, cot_out, st);
  else bwi_f16_variants<24>(L, a, gy, wplb, ga, N, cot_in, wmax, cot_out, st);
}

template <int C1, int FE>
static void bww_f16_launch(const LayerView& L, const float* a, const float* gy, float* part, int64_t N16, int64_t chunk,
                           int nx, int ny, int nmain, const unsigned* act_in, const unsigned* cot_in, hipStream_t st) {
  auto kb = (L.act == kActSiLU) ? bwd_weight_f16_kernel<C1, FE, kActSiLU, false>
                                : bwd_weight_f16_kernel<C1, FE, kActIdentity, false>;
  auto kg = (L.act == kActSiLU) ? bwd_weight_f16_kernel<C1, FE, kActSiLU, true>
                                : bwd_weight_f16_kernel<C1, FE, kActIdentity, true>;
  const size_t lds_b = (size_t)4 * (2 * FE * C1 * kWgRw + FE * kKnotPadU) * 4;
  const size_t lds_g = (size_t)4 * (2 * FE * C1 * kWgRw + FE * kKnotPadN) * 4;
  allow_lds(kb, lds_b);
  allow_lds(kg, lds_g);
  const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gy);
  kb<<<(unsigned)(nx * ny * nmain), 256, lds_b, st>>>(a4, g4, L.grid, L.ginfo, part, N16, L.I, L.O, chunk, nx, ny, nmain,
                                                      act_in, cot_in);
  kg<<<(unsigned)(nx * ny * nmain), 256, lds_g, st>>>(a4, g4, L.grid, L.ginfo, part, N16, L.I, L.O, chunk, nx, ny, nmain,
                                                      act_in, cot_in);
}

// Layers whose fp16 weight gradients run on 32 x 32 tiles (bwd_weight_f16w_kernel): blocks of 64 features at grid 10,
// of 32 at grid 20
bool f16_wide_weight_tiles(int D, int D2, const LayerView& L) {
  if (!f16_eligible(D, D2, L)) return false;
  if (!((L.G == 10 && L.I % 64 == 0) || (L.G == 20 && L.I % 32 == 0))) return false;
  const char* off = getenv("PDEKAN_DISABLE_F16_WIDE");   // A/B switch for tests and profiling
  return !(off && off[0] == '1');
}

template <int C1, int FE>
static void bww_f16w_launch(const LayerView& L, const float* a, const float* gy, float* part, int64_t N16, int64_t chunk,
                            int ny, int nmain, const unsigned* act_in, const unsigned* cot_in, hipStream_t st) {
  auto kb = (L.act == kActSiLU) ? bwd_weight_f16w_kernel<C1, FE, kActSiLU, false>
                                : bwd_weight_f16w_kernel<C1, FE, kActIdentity, false>;
  auto kg = (L.act == kActSiLU) ? bwd_weight_f16w_kernel<C1, FE, kActSiLU, true>
                                : bwd_weight_f16w_kernel<C1, FE, kActIdentity, true>;
  const size_t lds_b = (size_t)4 * WgW<C1, FE>::wave_words(false) * 4;
  const size_t lds_g = (size_t)4 * WgW<C1, FE>::wave_words(true) * 4;
  allow_lds(kb, lds_b);
  allow_lds(kg, lds_g);
  const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gy);
  const int nx = L.I / (4 * FE);
  kb<<<(unsigned)(nx * ny * nmain), 256, lds_b, st>>>(a4, g4, L.grid, L.ginfo, part, N16, L.I, L.O, chunk, nx, ny, nmain,
                                                      act_in, cot_in);
  kg<<<(unsigned)(nx * ny * nmain), 256, lds_g, st>>>(a4, g4, L.grid, L.ginfo, part, N16, L.I, L.O, chunk, nx, ny, nmain,
                                                      act_in, cot_in);
}

// The two variants on the grid of bwd_weight_mfma_kernel (blocks of 4*FE features x 64 outputs x nmain chunks of
// `chunk` points, N16 and chunk multiples of 16): a block runs in the variant that matches its features' knot rows.
// act_in / cot_in: per-jet maxima of a and gy (4 words each).
void launch_bwd_weight_f16_blocks(const LayerView& L, const float* a, const float* gy, float* part, int64_t N16,
                                  int64_t chunk, int nx, int ny, int nmain, const unsigned* act_in,
                                  const unsigned* cot_in, hipStream_t st) {
  if (f16_wide_weight_tiles(2, 1, L)) {
    if (L.G == 10) bww_f16w_launch<14, 16>(L, a, gy, part, N16, chunk, ny, nmain, act_in, cot_in, st);
    else bww_f16w_launch<24, 8>(L, a, gy, part, N16, chunk, ny, nmain, act_in, cot_in, st);
  } else if (L.G == 10) bww_f16_launch<14, 8>(L, a, gy, part, N16, chunk, nx, ny, nmain, act_in, cot_in, st);
  else bww_f16_launch<24, 4>(L, a, gy, part, N16, chunk, nx, ny, nmain, act_in, cot_in, st);
}

// 12-jet weight gradients on the grid of launch_bwd_weight_f16_blocks (grid 10); act_in / cot_in: per-slot maxima of a
// and gy (12 words each).  Writes nothing unless every knot row of the layer is (nearly) equally spaced.
void launch_bwd_weight12_f16(const LayerView& L, const float* a, const float* gy, float* part, int64_t N16, int64_t chunk,
                             int nx, int ny, int nmain, const unsigned* act_in, const unsigned* cot_in, hipStream_t st) {
  constexpr int C1 = 14, FE = 8;
  auto kb = (L.act == kActSiLU) ? bwd_weight12_f16_kernel<C1, FE, kActSiLU, false>
                                : bwd_weight12_f16_kernel<C1, FE, kActIdentity, false>;
  auto kg = (L.act == kActSiLU) ? bwd_weight12_f16_kernel<C1, FE, kActSiLU, true>
                                : bwd_weight12_f16_kernel<C1, FE, kActIdentity, true>;
  const size_t lds_b = (size_t)4 * (2 * FE * C1 * kWgRw + FE * kKnotPadU) * 4;
  const size_t lds_g = (size_t)4 * (2 * FE * C1 * kWgRw + FE * kKnotPadN) * 4;
  allow_lds(kb, lds_b);
  allow_lds(kg, lds_g);
  const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gy);
  kb<<<(unsigned)(nx * ny * nmain), 256, lds_b, st>>>(a4, g4, L.grid, L.ginfo, part, N16, L.I, L.O, chunk, nx, ny, nmain,
                                                      act_in, cot_in);
  kg<<<(unsigned)(nx * ny * nmain), 256, lds_g, st>>>(a4, g4, L.grid, L.ginfo, part, N16, L.I, L.O, chunk, nx, ny, nmain,
                                                      act_in, cot_in);
}

}  // namespace pdekan
