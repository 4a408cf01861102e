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
  if (D == 2) head_dispatch<2, 1>(L, a, x, sn, sd, target, wp, sigma, rate, inv_n, res, ga, part, loss_part, nblocks, N, st, cot_out);
  else head_dispatch<6, 5>(L, a, x, sn, sd, target, wp, sigma, rate, inv_n, res, ga, part, loss_part, nblocks, N, st, cot_out);
}

}  // namespace pdekan
