// Fused head of the Black-Scholes PINN step for gfx950: last KAN layer (one
// output) forward on jets + PDE residual + squared-error loss + the head's
// backward (input-jet gradients and packed weight gradients) in ONE pass over
// the head's input jets.  Replaces four launches and three re-reads of the
// widest activation tensor.  Reference computation: KANLinear.forward
// (efficient_kan/kan.py:153-166) of the last layer, residual_net / loss_res
// (run1.py:120-149) and their autograd backward (run1.py:225).
//
// A wave owns one point at a time; its 64 lanes are the head's input features
// (NF features per lane when in_features > 64).  Each lane evaluates its
// feature's basis once, the output jet is a wavefront-level butterfly sum, every
// lane then knows the residual and pulls the cotangent back through its own
// feature.  Weight-gradient partial sums live in a per-wave LDS table whose row
// i is touched by lane i only (plain read-modify-write, no atomics, so the
// result is bitwise reproducible); tables are added in wave order at the end.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kan_math.h"
#include "kan_launch.h"
#include "kan_wave.h"

namespace pdekan {

// LDS rows of the packed weights and of the gradient tables are PADDED: 3 columns in front of and behind the C
// spline columns (so the 4 live columns c0..c0+3, c0 in [-3, C-1] + what a row of knots allows, are always inside
// the row and need no masking: the pads hold zero weights and collect sums nobody reads), then the base column.
// Row I (all zero) serves the lanes past the last feature.
// Row lengths are odd: lane i works on row i, and rows an even number of words apart put lanes 16 (or 32) apart on
// the same LDS bank.
__host__ __device__ __forceinline__ int head_row_words(int C) { return (C + 7) | 1; }
__host__ __device__ __forceinline__ int head_knot_words(int nk) { return nk + 1; }   // (odd here too would cost the fifth block per compute unit at 64 inputs)

// ACT: the layer's base activation, a compile-time constant (a run-time kind costs the SiLU arithmetic on every
// point of an identity-activation head, selected away afterwards)
template <int D, int D2, int NF, int ACT>
__global__ void __launch_bounds__(256)
head_bs_kernel(const float* __restrict__ a, const float* __restrict__ x, int64_t stride_n, int64_t stride_d,
               const float* __restrict__ target, const float* __restrict__ grid, const float* __restrict__ ginfo,
               const float* __restrict__ wp,
               float sigma, float rate, float inv_n, float* __restrict__ res_out, float* __restrict__ ga,
               float* __restrict__ part, float* __restrict__ loss_part, int64_t N, int I, int G,
               unsigned* __restrict__ cot_out) {
  constexpr int J = 1 + D + D2;
  static_assert(D2 == D - 1, "assets first, time last");
  static_assert(J % 4 == 0, "output jets are summed over the wave four at a time");
  const int nk = G + 2 * kOrder + 1, C = G + kOrder, C1 = C + 1, CP = head_row_words(C);
  const int nw = blockDim.x >> 6;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int kr = head_knot_words(nk);                     // knots + the feature's 1/h (0 = not equally spaced) [+ pad]
  const int tab = (I + 1) * CP;
  float* s_knots = reinterpret_cast<float*>(smem);        // [I][kr]
  float* s_w = s_knots + I * kr;                          // [I+1][CP]
  float* s_acc = s_w + tab;                               // [nw][I+1][CP]
  float* s_loss = s_acc + nw * tab;                       // [nw]

  for (int t = threadIdx.x; t < I * kr; t += blockDim.x) {
    const int f = t / kr, k = t - f * kr;
    s_knots[t] = (k < nk) ? grid[(size_t)f * nk + k] : ((k == nk && ginfo) ? ginfo[f] : 0.0f);
  }
  for (int t = threadIdx.x; t < tab; t += blockDim.x) {
    const int i = t / CP, k = t - i * CP;
    float v = 0.0f;
    if (i < I) {
      if (k >= 3 && k < C + 3) v = wp[i * C1 + k - 3];
      else if (k == C + 6) v = wp[i * C1 + C];
    }
    s_w[t] = v;
  }
  for (int t = threadIdx.x; t < nw * tab; t += blockDim.x) s_acc[t] = 0.0f;
  __syncthreads();

  // the wave index, and with it the point a wave works on, is kept in scalar registers: row addresses are then
  // scalar arithmetic and the per-lane part of every address is a constant offset
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float hs2 = 0.5f * sigma * sigma;
  float loss_local = 0.0f;
  float gmax[J];                                          // per-jet maxima of the cotangents written to ga
#pragma unroll
  for (int j = 0; j < J; ++j) gmax[j] = 0.f;

  // what a lane keeps about its NF features for the whole kernel
  const float* krow[NF];
  const float* wrow[NF];
  float* arow[NF];
  float g0[NF], glast[NF], invh[NF], wbase[NF];
  int joff[NF];
  bool live[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int i = lane + 64 * f;
    live[f] = i < I;
    const int ic = live[f] ? i : 0, iw = live[f] ? i : I;
    krow[f] = s_knots + ic * kr;
    g0[f] = krow[f][0]; glast[f] = krow[f][nk - 1]; invh[f] = krow[f][nk];
    wrow[f] = s_w + iw * CP + 3;
    arow[f] = s_acc + wave * tab + iw * CP + 3;
    wbase[f] = wrow[f][C + 3];
    joff[f] = ic * J;
  }

  // A wave works on one point at a time, so the loads of the NEXT point (its jets, coordinates and
  // target) are issued before the current point's arithmetic: their HBM latency hides behind it.
  const int64_t pstride = (int64_t)gridDim.x * nw;
  const int64_t pfirst = (int64_t)blockIdx.x * nw + wave;
  float jetn[NF][J], xsn[D2], tgtn = 0.0f;
#define PDEKAN_HEAD_FETCH(PP)                                                          \
  {                                                                                    \
    const int64_t pp_ = ((PP) < N) ? (PP) : N - 1;                                     \
    const float* src_ = a + pp_ * I * J;                                               \
    _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                   \
      _Pragma("unroll") for (int j = 0; j < J; ++j) jetn[f][j] = src_[joff[f] + j];    \
    }                                                                                  \
    _Pragma("unroll") for (int d = 0; d < D2; ++d) xsn[d] = x[pp_ * stride_n + d * stride_d]; \
    tgtn = target ? target[pp_] : 0.0f;                                                \
  }
  PDEKAN_HEAD_FETCH(pfirst)

  for (int64_t p = pfirst; p < N; p += pstride) {
    float jet[NF][J];
    float xs[D2];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int j = 0; j < J; ++j) jet[f][j] = jetn[f][j];
#pragma unroll
    for (int d = 0; d < D2; ++d) xs[d] = xsn[d];
    const float tgt = tgtn;
    PDEKAN_HEAD_FETCH(p + pstride)
    Basis bs[NF];
    Act av[NF];
    float ph1[NF], ph2[NF], ph3[NF];
    float y[J];
#pragma unroll
    for (int j = 0; j < J; ++j) y[j] = 0.0f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (invh[f] > 0.0f) eval_basis_uniform_oneshot<true>(krow[f], nk, g0[f], glast[f], invh[f], jet[f][0], bs[f]);
      else eval_basis<true>(krow[f], nk, jet[f][0], bs[f]);
      av[f] = eval_act(ACT, jet[f][0]);
      const float* w = wrow[f] + bs[f].c0;
      const float wb = wbase[f];
      float p0 = wb * av[f].f0, p1 = wb * av[f].f1, p2 = wb * av[f].f2, p3 = wb * av[f].f3;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float wm = w[m];
        p0 += wm * bs[f].b[m]; p1 += wm * bs[f].d1[m]; p2 += wm * bs[f].d2[m]; p3 += wm * bs[f].d3[m];
      }
      ph1[f] = p1; ph2[f] = p2; ph3[f] = p3;
      jet_accumulate<D, D2>(y, jet[f], p0, p1, p2);
    }
    // wavefront sums on the VALU (DPP + permlane swaps), four jets at a time; the results are wave-uniform
#pragma unroll
    for (int j = 0; j < J; j += 4) wave_sum4(y[j], y[j + 1], y[j + 2], y[j + 3]);
    float res = y[D] + rate * y[0];
#pragma unroll
    for (int d = 0; d < D2; ++d) res = res - hs2 * xs[d] * xs[d] * y[1 + D + d] - rate * xs[d] * y[1 + d];
    const float diff = res - tgt;
    loss_local += diff * diff;                            // the same in every lane
    if (lane == 0 && res_out) res_out[p] = res;
    if (part || ga) {
      const float g = 2.0f * diff * inv_n;
      float gy[J];
      gy[0] = g * rate;
#pragma unroll
      for (int d = 0; d < D2; ++d) { gy[1 + d] = -g * rate * xs[d]; gy[1 + D + d] = -g * hs2 * xs[d] * xs[d]; }
      gy[D] = g;
      float* dst_row = ga + p * I * J;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        float A0, A1, A2;
        jet_pullback_coeffs<D, D2>(gy, jet[f], A0, A1, A2);
        // (lanes past the last feature add into the spare row; plain read-add-write: the LDS unit's own float add,
        // ds_add_f32, measured 4x slower for the whole kernel)
        if (part) {
          float* row = arow[f] + bs[f].c0;
#pragma unroll
          for (int m = 0; m < 4; ++m) row[m] += A0 * bs[f].b[m] + A1 * bs[f].d1[m] + A2 * bs[f].d2[m];
          arow[f][C + 3] += A0 * av[f].f0 + A1 * av[f].f1 + A2 * av[f].f2;
        }
        if (ga && live[f]) {
          float out[J];
          out[0] = A0 * ph1[f] + A1 * ph2[f] + A2 * ph3[f];
#pragma unroll
          for (int d = 0; d < D; ++d) {
            float v = gy[1 + d] * ph1[f];
            if (d < D2) v += 2.0f * jet[f][1 + d] * gy[1 + D + d] * ph2[f];
            out[1 + d] = v;
          }
#pragma unroll
          for (int d = 0; d < D2; ++d) out[1 + D + d] = gy[1 + D + d] * ph1[f];
#pragma unroll
          for (int j = 0; j < J; ++j) dst_row[joff[f] + j] = out[j];
#pragma unroll
          for (int j = 0; j < J; ++j) absmax_into(gmax[j], out[j]);
        }
      }
    }
  }
#undef PDEKAN_HEAD_FETCH
  if (cot_out && ga) publish_slot_absmax<J>(gmax, cot_out);
  if (lane == 0) s_loss[wave] = loss_local;
  __syncthreads();
  if (part) {
    float* dst = part + (size_t)blockIdx.x * I * C1;
    for (int t = threadIdx.x; t < I * C1; t += blockDim.x) {
      const int i = t / C1, c = t - i * C1;
      const int src = i * CP + (c < C ? c + 3 : C + 6);
      float s = 0.0f;
      for (int w = 0; w < nw; ++w) s += s_acc[w * tab + src];
      dst[t] = s;
    }
  }
  if (threadIdx.x == 0) {
    float s = 0.0f;
    for (int w = 0; w < nw; ++w) s += s_loss[w];
    loss_part[blockIdx.x] = s;
  }
}

// The same for heads whose gradient table is too large to give every wave its own (256 inputs at grid 20: 32 KB a
// table, three waves per compute unit with the kernel above): the I / 64 waves of a GROUP take 64 input features each
// of the SAME point and share one table (a row is still touched by one lane only); a block holds as many groups as
// tables fit beside the knot rows and the (unpadded) weight rows.  The output jet of a point is the sum of its
// waves' sums, exchanged through LDS with one block barrier per point (partial jets double-buffered); every wave of
// the group then knows the residual and pulls the cotangent back through its own features.  All waves of a block run
// the same number of iterations (iters); points past the end contribute zeros.
__host__ __device__ __forceinline__ int head_split_weight_words(int C1) { return C1 | 1; }
// knot rows of that kernel: the first knot three more times in front, the last four more times behind (what
// eval_basis_near_uniform reads without index clamps), then the feature's spacing word; an odd number of words
__host__ __device__ __forceinline__ int head_split_knot_words(int nk) { return (nk + 8) | 1; }

constexpr int kHeadSplitPoints = 3;          // points a group of waves works on per iteration

template <int D, int D2, int ACT>
__global__ void __launch_bounds__(768)
head_bs_split_kernel(const float* __restrict__ a, const float* __restrict__ x, int64_t stride_n, int64_t stride_d,
                     const float* __restrict__ target, const float* __restrict__ grid, const float* __restrict__ ginfo,
                     const float* __restrict__ wp, float sigma, float rate, float inv_n, float* __restrict__ res_out,
                     float* __restrict__ ga, float* __restrict__ part, float* __restrict__ loss_part, int64_t N, int I,
                     int G, unsigned* __restrict__ cot_out, int iters) {
  constexpr int J = 1 + D + D2;
  static_assert(D2 == D - 1, "assets first, time last");
  static_assert(J == 4, "one float4 per jet");
  const int nk = G + 2 * kOrder + 1, C = G + kOrder, C1 = C + 1, CP = head_row_words(C);
  const int WR = head_split_weight_words(C1);
  const int nw = blockDim.x >> 6, wpg = I >> 6, ng = nw / wpg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int kr = head_split_knot_words(nk);
  const int tab = I * CP;
  float* s_knots = reinterpret_cast<float*>(smem);        // [I][kr]: padded knot row, spacing word at nk + 7
  float* s_w = s_knots + I * kr;                          // [I][WR]: spline columns, base column at C
  float* s_acc = s_w + ((I * WR + 3) & ~3);               // [ng][I][CP]
  float* s_y = s_acc + ng * tab;                          // [2][points per iteration][nw][4]
  float* s_loss = s_y + 2 * kHeadSplitPoints * nw * 4;    // [nw]

  for (int t = threadIdx.x; t < I * kr; t += blockDim.x) {
    const int f = t / kr, k = t - f * kr;
    const int kk = k < 3 ? 0 : (k - 3 < nk ? k - 3 : nk - 1);
    s_knots[t] = (k < nk + 7) ? grid[(size_t)f * nk + kk] : ((k == nk + 7 && ginfo) ? ginfo[f] : 0.0f);
  }
  for (int t = threadIdx.x; t < I * WR; t += blockDim.x) {
    const int i = t / WR, k = t - i * WR;
    s_w[t] = (k < C1) ? wp[i * C1 + k] : 0.0f;
  }
  for (int t = threadIdx.x; t < ng * tab; t += blockDim.x) s_acc[t] = 0.0f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave / wpg, q = wave - grp * wpg;
  const int i = q * 64 + lane;                            // the lane's input feature, for the whole kernel
  const float hs2 = 0.5f * sigma * sigma;
  float loss_local = 0.0f;
  float gmax[J];
#pragma unroll
  for (int j = 0; j < J; ++j) gmax[j] = 0.f;
  const float* krow = s_knots + i * kr + 3;               // knot 0
  const float g0 = krow[0], glast = krow[nk - 1], invh = krow[nk + 4];
  const float* wrow = s_w + i * WR;
  const float wbase = wrow[C];
  float* arow = s_acc + grp * tab + i * CP + 3;
  const int joff = i * J;

  // A group works on PP points per iteration (their chains of LDS round trips overlap; one block barrier for the
  // PP of them); the loads of an iteration's points (jets, coordinates, targets) are issued one iteration ahead.
  constexpr int PP = kHeadSplitPoints;
  const int64_t pstride = (int64_t)gridDim.x * ng;
  const int64_t pfirst = (int64_t)blockIdx.x * ng + grp;
  struct Fetched { float4 jet; float xs[D2]; float tgt; };
  auto fetch = [&](int64_t pp) {
    const int64_t pc = pp < N ? pp : N - 1;
    Fetched f;
    f.jet = *reinterpret_cast<const float4*>(a + pc * I * J + joff);
#pragma unroll
    for (int d = 0; d < D2; ++d) f.xs[d] = x[pc * stride_n + d * stride_d];
    f.tgt = target ? target[pc] : 0.0f;
    return f;
  };
  Fetched nx[PP];
#pragma unroll
  for (int k = 0; k < PP; ++k) nx[k] = fetch(pfirst + k * pstride);

  for (int it = 0; it < iters; ++it) {
    int64_t p[PP];
    bool valid[PP];
    float jet[PP][J], xs[PP][D2], tgt[PP];
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      p[k] = pfirst + ((int64_t)it * PP + k) * pstride;
      valid[k] = p[k] < N;
      jet[k][0] = nx[k].jet.x; jet[k][1] = nx[k].jet.y; jet[k][2] = nx[k].jet.z; jet[k][3] = nx[k].jet.w;
#pragma unroll
      for (int d = 0; d < D2; ++d) xs[k][d] = nx[k].xs[d];
      tgt[k] = nx[k].tgt;
    }
#pragma unroll
    for (int k = 0; k < PP; ++k) nx[k] = fetch(p[k] + PP * pstride);
    Basis bs[PP];
    Act av[PP];
    float p1[PP], p2[PP], p3[PP];
    float* sy = s_y + (it & 1) * PP * nw * 4;
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      if (invh > 0.0f) eval_basis_uniform_oneshot<true>(krow, nk, g0, glast, invh, jet[k][0], bs[k]);
      else if (invh < 0.0f) eval_basis_near_uniform<true>(krow, nk, g0, glast, -invh, jet[k][0], bs[k]);
      else eval_basis<true>(krow, nk, jet[k][0], bs[k]);
      av[k] = eval_act(ACT, jet[k][0]);
      float q0 = wbase * av[k].f0, q1 = wbase * av[k].f1, q2 = wbase * av[k].f2, q3 = wbase * av[k].f3;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int c = bs[k].c0 + m;
        const bool in = c >= 0 && c < C;
        const float wm = in ? wrow[in ? c : 0] : 0.0f;
        q0 += wm * bs[k].b[m]; q1 += wm * bs[k].d1[m]; q2 += wm * bs[k].d2[m]; q3 += wm * bs[k].d3[m];
      }
      p1[k] = q1; p2[k] = q2; p3[k] = q3;
      float y[J];
#pragma unroll
      for (int j = 0; j < J; ++j) y[j] = 0.0f;
      jet_accumulate<D, D2>(y, jet[k], q0, q1, q2);
      wave_sum4(y[0], y[1], y[2], y[3]);
      if (lane == 0) *reinterpret_cast<float4*>(sy + (k * nw + wave) * 4) = make_float4(y[0], y[1], y[2], y[3]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      float y[J];
#pragma unroll
      for (int j = 0; j < J; ++j) y[j] = 0.0f;
      for (int w = 0; w < wpg; ++w) {                     // the group's waves, in order
        const float4 v = *reinterpret_cast<const float4*>(sy + (k * nw + grp * wpg + w) * 4);
        y[0] += v.x; y[1] += v.y; y[2] += v.z; y[3] += v.w;
      }
      float res = y[D] + rate * y[0];
#pragma unroll
      for (int d = 0; d < D2; ++d) res = res - hs2 * xs[k][d] * xs[k][d] * y[1 + D + d] - rate * xs[k][d] * y[1 + d];
      const float diff = valid[k] ? res - tgt[k] : 0.0f;
      if (q == 0) loss_local += diff * diff;
      if (lane == 0 && q == 0 && valid[k] && res_out) res_out[p[k]] = res;
      if (part || ga) {
        const float g = 2.0f * diff * inv_n;
        float gy[J];
        gy[0] = g * rate;
#pragma unroll
        for (int d = 0; d < D2; ++d) { gy[1 + d] = -g * rate * xs[k][d]; gy[1 + D + d] = -g * hs2 * xs[k][d] * xs[k][d]; }
        gy[D] = g;
        float A0, A1, A2;
        jet_pullback_coeffs<D, D2>(gy, jet[k], A0, A1, A2);
        if (part) {                                       // (the PP points of a lane may share columns: in program order)
          float* row = arow + bs[k].c0;
#pragma unroll
          for (int m = 0; m < 4; ++m) row[m] += A0 * bs[k].b[m] + A1 * bs[k].d1[m] + A2 * bs[k].d2[m];
          arow[C + 3] += A0 * av[k].f0 + A1 * av[k].f1 + A2 * av[k].f2;
        }
        if (ga && valid[k]) {
          float out[J];
          out[0] = A0 * p1[k] + A1 * p2[k] + A2 * p3[k];
#pragma unroll
          for (int d = 0; d < D; ++d) {
            float v = gy[1 + d] * p1[k];
            if (d < D2) v += 2.0f * jet[k][1 + d] * gy[1 + D + d] * p2[k];
            out[1 + d] = v;
          }
#pragma unroll
          for (int d = 0; d < D2; ++d) out[1 + D + d] = gy[1 + D + d] * p1[k];
          *reinterpret_cast<float4*>(ga + p[k] * I * J + joff) = make_float4(out[0], out[1], out[2], out[3]);
#pragma unroll
          for (int j = 0; j < J; ++j) absmax_into(gmax[j], out[j]);
        }
      }
    }
  }
  if (cot_out && ga) publish_slot_absmax<J>(gmax, cot_out);
  if (lane == 0) s_loss[wave] = loss_local;
  __syncthreads();
  if (part) {
    float* dst = part + (size_t)blockIdx.x * I * C1;
    for (int t = threadIdx.x; t < I * C1; t += blockDim.x) {
      const int f = t / C1, c = t - f * C1;
      const int src = f * CP + (c < C ? c + 3 : C + 6);
      float s = 0.0f;
      for (int w = 0; w < ng; ++w) s += s_acc[w * tab + src];
      dst[t] = s;
    }
  }
  if (threadIdx.x == 0) {
    float s = 0.0f;
    for (int w = 0; w < nw; ++w) s += s_loss[w];
    loss_part[blockIdx.x] = s;
  }
}

static size_t head_table_bytes(const LayerView& L) { return (size_t)(L.I + 1) * head_row_words(L.C()) * 4; }

constexpr size_t kLdsPerBlockMax = 160 * 1024;            // gfx950: 160 KiB per workgroup

// waves per block: 4 where several blocks still fit a CU beside each other, else as many (up to 4) as one block's LDS
// holds - a head with 256 inputs at grid 20 keeps 3 gradient tables of 32 KB beside its weights and knots
static int head_waves(const LayerView& L) {
  const size_t table = head_table_bytes(L);
  if (table * 4 <= 64 * 1024) return 4;
  const size_t fixed = (size_t)L.I * head_knot_words(L.nk()) * 4 + table + 64;
  for (int nw = 4; nw > 1; --nw)
    if (fixed + table * nw <= kLdsPerBlockMax) return nw;
  return 1;
}

// groups of I / 64 waves that share a table (head_bs_split_kernel): as many as the LDS holds, 16 waves at most;
// 0 = the layer stays on the table-per-wave kernel (heads of one asset + time whose tables leave it under 4 waves)
constexpr size_t kHeadSplitWaveBytes = 2 * kHeadSplitPoints * 16 + 4;   // a wave's partial jets (double-buffered) and loss sum
static size_t head_split_fixed_bytes(const LayerView& L) {
  return (size_t)L.I * head_split_knot_words(L.nk()) * 4 + (((size_t)L.I * head_split_weight_words(L.C1()) + 3) & ~(size_t)3) * 4;
}
static int head_split_groups(int D, const LayerView& L) {
  const char* off = getenv("PDEKAN_DISABLE_HEAD_SPLIT");
  if ((off && off[0] == '1') || D != 2 || L.I < 128 || L.I % 64 != 0 || head_waves(L) >= 4) return 0;
  const int wpg = L.I / 64;
  const size_t table = (size_t)L.I * head_row_words(L.C()) * 4;
  int ng = 0;
  while ((ng + 1) * wpg <= 12 &&             // (768 threads: three waves per SIMD, 168 registers each)
         head_split_fixed_bytes(L) + (size_t)(ng + 1) * table + (size_t)(ng + 1) * wpg * kHeadSplitWaveBytes + 64 <= kLdsPerBlockMax) ++ng;
  return ng * wpg >= 8 ? ng : 0;              // (worth it from 8 waves per compute unit)
}
static size_t head_split_lds_bytes(const LayerView& L, int ng) {
  const int nw = ng * (L.I / 64);
  return head_split_fixed_bytes(L) + (size_t)ng * L.I * head_row_words(L.C()) * 4 + (size_t)nw * kHeadSplitWaveBytes;
}

// dynamic LDS of a block: knot rows (+ 1/h), the padded weight rows, one gradient table per wave, the waves' loss sums
static size_t head_lds_bytes(const LayerView& L) {
  const int nw = head_waves(L);
  return (size_t)L.I * head_knot_words(L.nk()) * 4 + head_table_bytes(L) * (1 + nw) + (size_t)nw * 4;
}

// Wide heads on fine grids (256 inputs at grid 48 and up) do not fit the LDS: those take the unfused
// last-layer kernels + bs_residual_kernel.
bool head_eligible(int D, const LayerView& L) {
  return (D == 2 || D == 6) && L.O == 1 && L.I <= 256 && head_lds_bytes(L) <= kLdsPerBlockMax;
}

int head_blocks(int64_t N, const LayerView& L) {
  if (const int ng = head_split_groups(2, L)) {            // (asked for heads of one asset + time only: see launch_head_bs)
    int64_t b = (N + (int64_t)ng * 4 - 1) / ((int64_t)ng * 4);
    if (b > 1024) b = 1024;                               // one block per compute unit at a time, four rounds
    return (int)(b < 1 ? 1 : b);
  }
  const int nw = head_waves(L);
  // at least 4 points per wave (small batches are latency-bound: more, shorter blocks finish sooner)
  int64_t b = (N + (int64_t)nw * 4 - 1) / ((int64_t)nw * 4);
  if (b > kHeadBlocksMax) b = kHeadBlocksMax;             // 5 blocks per CU: the register budget's limit
  if (b < 1) b = 1;
  return (int)b;
}

template <int D, int D2, int NF, int ACT>
static void head_launch(const LayerView& L, const float* a, const float* x, int64_t sn, int64_t sd,
                        const float* target, const float* wp, float sigma, float rate, float inv_n, float* res,
                        float* ga, float* part, float* loss_part, int nblocks, int64_t N, hipStream_t st, unsigned* cot_out) {
  const int nw = head_waves(L);
  const size_t lds = head_lds_bytes(L);
  auto k = head_bs_kernel<D, D2, NF, ACT>;
  if (lds > 48 * 1024)
    ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
  k<<<nblocks, 64 * nw, lds, st>>>(a, x, sn, sd, target, L.grid, L.ginfo, wp, sigma, rate, inv_n, res, ga, part, loss_part,
                                   N, L.I, L.G, cot_out);
}

template <int D, int D2, int NF>
static void head_act(const LayerView& L, const float* a, const float* x, int64_t sn, int64_t sd,
                     const float* target, const float* wp, float sigma, float rate, float inv_n, float* res,
                     float* ga, float* part, float* loss_part, int nblocks, int64_t N, hipStream_t st, unsigned* cot_out) {
  if (L.act == kActSiLU) head_launch<D, D2, NF, kActSiLU>(L, a, x, sn, sd, target, wp, sigma, rate, inv_n, res, ga, part, loss_part, nblocks, N, st, cot_out);
  else head_launch<D, D2, NF, kActIdentity>(L, a, x, sn, sd, target, wp, sigma, rate, inv_n, res, ga, part, loss_part, nblocks, N, st, cot_out);
}

template <int D, int D2>
static void head_dispatch(const LayerView& L, const float* a, const float* x, int64_t sn, int64_t sd,
                          const float* target, const float* wp, float sigma, float rate, float inv_n, float* res,
                          float* ga, float* part, float* loss_part, int nblocks, int64_t N, hipStream_t st, unsigned* cot_out) {
  if (L.I <= 64) head_act<D, D2, 1>(L, a, x, sn, sd, target, wp, sigma, rate, inv_n, res, ga, part, loss_part, nblocks, N, st, cot_out);
  else if (L.I <= 128) head_act<D, D2, 2>(L, a, x, sn, sd, target, wp, sigma, rate, inv_n, res, ga, part, loss_part, nblocks, N, st, cot_out);
  else head_act<D, D2, 4>(L, a, x, sn, sd, target, wp, sigma, rate, inv_n, res, ga, part, loss_part, nblocks, N, st, cot_out);
}

void launch_head_bs(int D, const LayerView& L, const float* a, const float* x, int64_t sn, int64_t sd,
                    const float* target, const float* wp, float sigma, float rate, float inv_n, float* res,
                    float* ga, float* part, float* loss_part, int nblocks, int64_t N, hipStream_t st, unsigned* cot_out) {
  if (const int ng = (D == 2) ? head_split_groups(D, L) : 0) {
    const int nw = ng * (L.I / 64);
    const size_t lds = head_split_lds_bytes(L, ng);
    const int64_t per_iter = (int64_t)nblocks * ng * kHeadSplitPoints;
    const int iters = (int)((N + per_iter - 1) / per_iter);
    auto go = [&](auto k) {
      ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
      k<<<nblocks, 64 * nw, lds, st>>>(a, x, sn, sd, target, L.grid, L.ginfo, wp, sigma, rate, inv_n, res, ga, part, loss_part,
                                       N, L.I, L.G, cot_out, iters);
    };
    if (L.act == kActSiLU) go(head_bs_split_kernel<2, 1, kActSiLU>);
    else go(head_bs_split_kernel<2, 1, kActIdentity>);
    return;
  }
  if (D == 2) head_dispatch<2, 1>(L, a, x, sn, sd, target, wp, sigma, rate, inv_n, res, ga, part, loss_part, nblocks, N, st, cot_out);
  else head_dispatch<6, 5>(L, a, x, sn, sd, target, wp, sigma, rate, inv_n, res, ga, part, loss_part, nblocks, N, st, cot_out);
}

}  // namespace pdekan
