// Streaming (non-MFMA) kernels of the KAN layer for gfx950: used for layers
// whose contraction is too skinny to be a dense GEMM (first layer: few inputs;
// head: one output) and as the general path for any shape.
//
// Layouts (all fp32, row-major, innermost last):
//   jets    a[N][I][J], y[N][O][J]      J = 1 + D + D2   (see kan_math.h)
//   knots   grid[I][nk]                 nk = G + 2*3 + 1 (kan.py:27-36)
//   packed weights of one layer, C1 = G+3+1 columns per (o,i) edge:
//     wp[O][I][C1] : c < C1-1 -> spline_weight*spline_scaler (kan.py:145-151),
//                    c = C1-1 -> base_weight
//     wt[I*C1][O]  : the transpose, k = i*C1 + c
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kan_math.h"
#include "kan_launch.h"
#include "kan_wave.h"

namespace pdekan {

// One record per (point, input feature): everything phase 2 needs.
struct __attribute__((aligned(16))) EdgeRec {
  float b[4], d1[4], d2[4];   // bases and derivatives of the 4 live splines
  float f0, f1, f2;           // base activation and derivatives
  int c0;                     // first live spline index (may be out of range)
};

// Stage knot rows [nrows][nk] as LDS rows of nk+1 floats; the extra float is the
// feature's 1/h (equally spaced knots) or 0.
__device__ __forceinline__ void load_knots_to_lds(float* s_knots, const float* grid, const float* ginfo,
                                                  int row0, int nrows, int I, int nk) {
  const int kr = nk + 1;
  for (int t = threadIdx.x; t < nrows * kr; t += blockDim.x) {
    const int f = t / kr, k = t - f * kr;
    const int fi = row0 + f;
    float v = 0.0f;
    if (fi < I) v = (k < nk) ? grid[(size_t)fi * nk + k] : (ginfo ? ginfo[fi] : 0.0f);
    s_knots[t] = v;
  }
}

// per feature: 1/h when its knots are equally spaced, -1/h when nearly so, else 0 (see knot_spacing_info)
// (e.g. a grid moved by update_grid, kan.py:168-215)
__global__ void grid_info_kernel(const float* __restrict__ grid, float* __restrict__ ginfo, int I, int nk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < I) ginfo[i] = knot_spacing_info(grid + (size_t)i * nk, nk);
}

// --------------------------------------------------------------------------
// pack: parameters -> wp / wt
// --------------------------------------------------------------------------
// ginfo (or nullptr): the layer's knot-row classes are written by the same launch (what grid_info_kernel does)
__global__ void pack_weights_kernel(const float* __restrict__ base_w, const float* __restrict__ spline_w,
                                    const float* __restrict__ scaler, float* __restrict__ wp,
                                    float* __restrict__ wt, int I, int O, int C, const float* __restrict__ grid,
                                    float* __restrict__ ginfo, int nk) {
  const int C1 = C + 1;
  const int total = O * I * C1;
  if (ginfo)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < I; i += gridDim.x * blockDim.x)
      ginfo[i] = knot_spacing_info(grid + (size_t)i * nk, nk);
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int c = t % C1;
    const int oi = t / C1;
    const int i = oi % I, o = oi / I;
    float v;
    if (c < C) {
      v = spline_w[(size_t)oi * C + c];
      if (scaler) v *= scaler[oi];
    } else {
      v = base_w[oi];
    }
    wp[t] = v;
    wt[(size_t)(i * C1 + c) * O + o] = v;
  }
}

// The same for up to kPackLayersMax layers of a model in ONE launch (blockIdx.y = layer), which also clears the
// words at `clear` (the maxima the fp16 matrix kernels combine into): a step on 10^4 points is the sum of its launch
// latencies, and every layer's pack used to be two launches plus a memset.
struct PackLayer {
  const float* base_w; const float* spline_w; const float* scaler; const float* grid;
  float* wp; float* wt; float* ginfo;
  int I, O, C, nk;
};
struct PackModelArgs {
  PackLayer layer[kPackLayersMax];
  unsigned* clear; int clear_words;
};

__global__ void pack_model_kernel(PackModelArgs args) {
  const PackLayer L = args.layer[blockIdx.y];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  if (blockIdx.y == 0)
    for (int t = tid; t < args.clear_words; t += nth) args.clear[t] = 0u;
  for (int i = tid; i < L.I; i += nth) L.ginfo[i] = knot_spacing_info(L.grid + (size_t)i * L.nk, L.nk);
  const int C1 = L.C + 1;
  const int total = L.O * L.I * C1;
  for (int t = tid; t < total; t += nth) {
    const int c = t % C1;
    const int oi = t / C1;
    const int i = oi % L.I, o = oi / L.I;
    float v;
    if (c < L.C) {
      v = L.spline_w[(size_t)oi * L.C + c];
      if (L.scaler) v *= L.scaler[oi];
    } else {
      v = L.base_w[oi];
    }
    L.wp[t] = v;
    L.wt[(size_t)(i * C1 + c) * L.O + o] = v;
  }
}

// --------------------------------------------------------------------------
// seed: points x[N][d] (any strides) -> input jets a[N][d][J]
// --------------------------------------------------------------------------
template <int D, int D2>
__global__ void seed_jets_kernel(const float* __restrict__ x, int64_t stride_n, int64_t stride_d,
                                 float* __restrict__ a, int64_t N, int I) {
  constexpr int J = 1 + D + D2;
  const int64_t total = N * I;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / I;
    const int i = (int)(t % I);
    float* out = a + t * J;
    out[0] = x[p * stride_n + i * stride_d];
#pragma unroll
    for (int d = 0; d < D; ++d) out[1 + d] = (d == i) ? 1.0f : 0.0f;
#pragma unroll
    for (int d = 0; d < D2; ++d) out[1 + D + d] = 0.0f;
  }
}

// --------------------------------------------------------------------------
// forward, any shape.  A block owns PT points.  Phase 1: one thread per
// (point, input) evaluates the basis once and parks it in LDS.  Phase 2: one
// thread per (point, output, i-slice) walks the inputs, gathers the 4+1 live
// weights (coalesced over o through wt) and accumulates the output jet; the
// NIC i-slices of one output sit in adjacent lanes and are summed by shuffles.
// A launch covers the input features [i0, i0 + IB) (LDS holds that window only);
// wide layers are walked window by window, later windows adding into y.
// --------------------------------------------------------------------------
template <int D, int D2>
__global__ void __launch_bounds__(256)
fwd_generic_kernel(const float* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
                   const float* __restrict__ wt, float* __restrict__ y,
                   int64_t N, int I, int O, int G, int act, int PT, int NIC, int i0, int IB, int accumulate,
                   int skip_near_uniform, unsigned* __restrict__ amax_out, SeedSrc seed) {
  constexpr int J = 1 + D + D2;
  if (skip_near_uniform && all_rows_near_uniform(ginfo, I)) return;   // a matrix kernel beside this one takes the layer
  const int nk = G + 2 * kOrder + 1, C = G + kOrder, C1 = C + 1;
  const int IBe = (I - i0 < IB) ? (I - i0) : IB;          // features of this window
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EdgeRec* recs = reinterpret_cast<EdgeRec*>(smem);
  float* s_jet = reinterpret_cast<float*>(recs + PT * IB);
  float* s_knots = s_jet + PT * IB * J;
  load_knots_to_lds(s_knots, grid, ginfo, i0, IBe, I, nk);
  __syncthreads();

  const int64_t p0 = (int64_t)blockIdx.x * PT;
  const int npts = (int)((N - p0 < PT) ? (N - p0) : PT);

  for (int pair = threadIdx.x; pair < npts * IBe; pair += blockDim.x) {
    const int i = pair % IBe, pl = pair / IBe;
    float jet[J];
    if (seed.x) {                             // first layer of a model: the jets are built here and left for the backward
      float* dst = seed.a_out + ((p0 + pl) * I + i0 + i) * J;
      jet[0] = seed.x[(p0 + pl) * seed.sn + (i0 + i) * seed.sd];
#pragma unroll
      for (int d = 0; d < D; ++d) jet[1 + d] = (d == i0 + i) ? 1.0f : 0.0f;
#pragma unroll
      for (int d = 0; d < D2; ++d) jet[1 + D + d] = 0.0f;
#pragma unroll
      for (int j = 0; j < J; ++j) dst[j] = jet[j];
    } else {
      const float* src = a + ((p0 + pl) * I + i0 + i) * J;
#pragma unroll
      for (int j = 0; j < J; ++j) jet[j] = src[j];
    }
#pragma unroll
    for (int j = 0; j < J; ++j) s_jet[pair * J + j] = jet[j];
    Basis bs;
    eval_basis_row<false>(s_knots + i * (nk + 1), nk, jet[0], bs);
    const Act av = eval_act(act, jet[0]);
    EdgeRec r;
#pragma unroll
    for (int m = 0; m < 4; ++m) { r.b[m] = bs.b[m]; r.d1[m] = bs.d1[m]; r.d2[m] = bs.d2[m]; }
    r.f0 = av.f0; r.f1 = av.f1; r.f2 = av.f2; r.c0 = bs.c0;
    recs[pair] = r;
  }
  __syncthreads();

  float amax[J];                              // per-slot maxima of the finished outputs (launch of the last window)
#pragma unroll
  for (int j = 0; j < J; ++j) amax[j] = 0.0f;
  const int items = npts * O * NIC;
  const int rounds = (PT * O * NIC + blockDim.x - 1) / blockDim.x;
  for (int rd = 0; rd < rounds; ++rd) {
    const int t = rd * blockDim.x + threadIdx.x;
    const bool live = t < items;
    const int ic = t % NIC;
    const int po = t / NIC;
    const int o = po % O, p = po / O;
    float acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = 0.0f;
    if (live) {
      for (int i = ic; i < IBe; i += NIC) {
        const EdgeRec r = recs[p * IBe + i];
        const float* jet = s_jet + (p * IBe + i) * J;
        const float* wcol = wt + (size_t)((i0 + i) * C1) * O + o;
        const float wb = wcol[(size_t)C * O];
        float p0v = wb * r.f0, p1v = wb * r.f1, p2v = wb * r.f2;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int c = r.c0 + m;
          if (c >= 0 && c < C) {
            const float w = wcol[(size_t)c * O];
            p0v += w * r.b[m]; p1v += w * r.d1[m]; p2v += w * r.d2[m];
          }
        }
        jet_accumulate<D, D2>(acc, jet, p0v, p1v, p2v);
      }
    }
    for (int off = NIC >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int j = 0; j < J; ++j) acc[j] += __shfl_xor(acc[j], off);
    }
    if (live && ic == 0) {
      float* dst = y + ((p0 + p) * O + o) * J;
      if constexpr (J % 4 == 0) {               // whole jets as 16-byte accesses (a jet of 4 or 12 floats is 16-byte aligned)
        float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
        for (int q = 0; q < J / 4; ++q) {
          float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
          if (accumulate) {
            const float4 old = d4[q];
            v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
          }
          d4[q] = v;
          amax[4 * q] = fmaxf(amax[4 * q], fabsf(v.x));
          amax[4 * q + 1] = fmaxf(amax[4 * q + 1], fabsf(v.y));
          amax[4 * q + 2] = fmaxf(amax[4 * q + 2], fabsf(v.z));
          amax[4 * q + 3] = fmaxf(amax[4 * q + 3], fabsf(v.w));
        }
      } else {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const float v = accumulate ? dst[j] + acc[j] : acc[j];
          dst[j] = v;
          amax[j] = fmaxf(amax[j], fabsf(v));
        }
      }
    }
  }
  if (amax_out) publish_slot_absmax<J>(amax, amax_out);
}

// --------------------------------------------------------------------------
// backward to the input jet, any shape.  One thread per (point, input):
//   Gm[m][j] = sum_o w[o][i][c0+m] * gy[p][o][j]   (m = 4 live splines + base)
// then the chain rule through B, B', B'', B''' (cubic: the fourth derivative
// vanishes, so three derivative orders close the recursion).
// --------------------------------------------------------------------------
template <int D, int D2>
__global__ void __launch_bounds__(256)
bwd_input_generic_kernel(const float* __restrict__ a, const float* __restrict__ gy,
                         const float* __restrict__ grid, const float* __restrict__ ginfo, const float* __restrict__ wp,
                         float* __restrict__ ga, int64_t N, int I, int O, int G, int act, int i0, int IB,
                         int skip_near_uniform) {
  constexpr int J = 1 + D + D2;
  if (skip_near_uniform && all_rows_near_uniform(ginfo, I)) return;   // a matrix kernel beside this one takes the layer
  const int nk = G + 2 * kOrder + 1, C = G + kOrder, C1 = C + 1;
  const int IBe = (I - i0 < IB) ? (I - i0) : IB;          // this launch covers features [i0, i0 + IBe)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_knots = reinterpret_cast<float*>(smem);
  load_knots_to_lds(s_knots, grid, ginfo, i0, IBe, I, nk);
  __syncthreads();
  const int64_t total = N * IBe;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total;
       u += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = u / IBe;
    const int il = (int)(u % IBe);
    const int i = i0 + il;
    const int64_t t = p * I + i;
    float jet[J];
#pragma unroll
    for (int j = 0; j < J; ++j) jet[j] = a[t * J + j];
    Basis bs;
    eval_basis_row<true>(s_knots + il * (nk + 1), nk, jet[0], bs);
    const Act av = eval_act(act, jet[0]);
    float Gm[5][J];
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
      for (int j = 0; j < J; ++j) Gm[m][j] = 0.0f;
    bool ok[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) ok[m] = (bs.c0 + m >= 0) && (bs.c0 + m < C);
    const float* gyp = gy + p * O * J;
    for (int o = 0; o < O; ++o) {
      const float* w = wp + ((size_t)o * I + i) * C1;
      float wm[5];
#pragma unroll
      for (int m = 0; m < 4; ++m) wm[m] = ok[m] ? w[bs.c0 + m] : 0.0f;
      wm[4] = w[C];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const float g = gyp[o * J + j];
#pragma unroll
        for (int m = 0; m < 5; ++m) Gm[m][j] += wm[m] * g;
      }
    }
    float out[J];
#pragma unroll
    for (int j = 0; j < J; ++j) out[j] = 0.0f;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      const float e1 = (m < 4) ? bs.d1[m < 4 ? m : 0] : av.f1;
      const float e2 = (m < 4) ? bs.d2[m < 4 ? m : 0] : av.f2;
      const float e3 = (m < 4) ? bs.d3[m < 4 ? m : 0] : av.f3;
      out[0] += Gm[m][0] * e1;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        out[0] += Gm[m][1 + d] * jet[1 + d] * e2;
        out[1 + d] += Gm[m][1 + d] * e1;
      }
#pragma unroll
      for (int d = 0; d < D2; ++d) {
        const float gdd = Gm[m][1 + D + d];
        out[0] += gdd * (jet[1 + d] * jet[1 + d] * e3 + jet[1 + D + d] * e2);
        out[1 + d] += 2.0f * jet[1 + d] * gdd * e2;
        out[1 + D + d] += gdd * e1;
      }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) ga[t * J + j] = out[j];
  }
}

// --------------------------------------------------------------------------
// backward to the packed weights, any shape.  grid = (i-tiles, o-tiles, point
// chunks).  A block owns a TI x TO tile of edges and one chunk of points; its
// 256 threads are TE = TI*TO edges times PG point groups, and every thread
// keeps a private column of C1 partial sums in LDS (no atomics, fixed order:
// bitwise reproducible).  Points are taken PT at a time: phase 1 evaluates the
// basis once per (point, input of the tile); in phase 2 thread (edge, group g)
// folds in points g, g+PG, ...  The tile shape is chosen by the launcher so
// skinny layers (2 inputs, or 1 output) still use every thread.
// Output: part[chunk][O][I][C1].
// --------------------------------------------------------------------------
struct BwTile { int TI, TO, PG, PT; };

template <int D, int D2>
__global__ void __launch_bounds__(256)
bwd_weight_generic_kernel(const float* __restrict__ a, const float* __restrict__ gy,
                          const float* __restrict__ grid, const float* __restrict__ ginfo, float* __restrict__ part,
                          int64_t N, int I, int O, int G, int act, int64_t chunk_pts, BwTile T, int skip_near_uniform) {
  constexpr int J = 1 + D + D2;
  if (skip_near_uniform && all_rows_near_uniform(ginfo, I)) return;   // a matrix kernel beside this one takes the layer
  const int nk = G + 2 * kOrder + 1, C = G + kOrder, C1 = C + 1;
  const int TI = T.TI, TO = T.TO, PG = T.PG, PT = T.PT, TE = TI * TO;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EdgeRec* recs = reinterpret_cast<EdgeRec*>(smem);                  // [PT][TI]
  float* s_jet = reinterpret_cast<float*>(recs + PT * TI);           // [PT][TI][J]
  float* s_knots = s_jet + PT * TI * J;                              // [TI][nk+1]
  float* s_acc = s_knots + TI * (nk + 1);                            // [C1][256]

  const int i0 = blockIdx.x * TI, o0 = blockIdx.y * TO;
  const int e = threadIdx.x % TE, pg = threadIdx.x / TE;
  const int ti = e % TI, to = e / TI;
  const int i = i0 + ti, o = o0 + to;
  const bool edge_live = (i < I) && (o < O);

  load_knots_to_lds(s_knots, grid, ginfo, i0, TI, I, nk);
  for (int c = 0; c < C1; ++c) s_acc[c * 256 + threadIdx.x] = 0.0f;
  __syncthreads();

  const int64_t pbeg = (int64_t)blockIdx.z * chunk_pts;
  const int64_t pend = (pbeg + chunk_pts < N) ? pbeg + chunk_pts : N;
  for (int64_t pb = pbeg; pb < pend; pb += PT) {
    const int npts = (int)((pend - pb < PT) ? (pend - pb) : PT);
    for (int pair = threadIdx.x; pair < npts * TI; pair += blockDim.x) {
      const int pp = pair / TI, t1 = pair - pp * TI;
      const int fi = i0 + t1;
      if (fi < I) {
        const float* src = a + ((pb + pp) * I + fi) * J;
        float jet[J];
#pragma unroll
        for (int j = 0; j < J; ++j) { jet[j] = src[j]; s_jet[pair * J + j] = jet[j]; }
        Basis bs;
        eval_basis_row<false>(s_knots + t1 * (nk + 1), nk, jet[0], bs);
        const Act av = eval_act(act, jet[0]);
        EdgeRec r;
#pragma unroll
        for (int m = 0; m < 4; ++m) { r.b[m] = bs.b[m]; r.d1[m] = bs.d1[m]; r.d2[m] = bs.d2[m]; }
        r.f0 = av.f0; r.f1 = av.f1; r.f2 = av.f2; r.c0 = bs.c0;
        recs[pair] = r;
      }
    }
    __syncthreads();
    if (edge_live) {
      for (int pp = pg; pp < npts; pp += PG) {
        const EdgeRec r = recs[pp * TI + ti];
        float g[J];
        const float* gsrc = gy + ((pb + pp) * O + o) * J;
        if constexpr (J % 4 == 0) {             // whole jets as 16-byte loads
#pragma unroll
          for (int q = 0; q < J / 4; ++q) {
            const float4 v = reinterpret_cast<const float4*>(gsrc)[q];
            g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; g[4 * q + 3] = v.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < J; ++j) g[j] = gsrc[j];
        }
        float A0, A1, A2;
        jet_pullback_coeffs<D, D2>(g, s_jet + (pp * TI + ti) * J, A0, A1, A2);
        s_acc[C * 256 + threadIdx.x] += A0 * r.f0 + A1 * r.f1 + A2 * r.f2;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int c = r.c0 + m;
          if (c >= 0 && c < C) s_acc[c * 256 + threadIdx.x] += A0 * r.b[m] + A1 * r.d1[m] + A2 * r.d2[m];
        }
      }
    }
    __syncthreads();
  }
  if (edge_live && pg == 0) {
    float* dst = part + (((size_t)blockIdx.z * O + o) * I + i) * C1;
    for (int c = 0; c < C1; ++c) {
      float sum = 0.0f;
      for (int g = 0; g < PG; ++g) sum += s_acc[c * 256 + e + TE * g];
      dst[c] = sum;
    }
  }
}

// ==========================================================================
// First layer of a model that carries many jets (the 5-asset basket: inputs S_1..S_5, t; D = 6 directions, D2 = 5
// second derivatives, 12 floats per value).  The layer's input jets are seeds - the value, 1 in the input's own
// direction, zeros elsewhere - so with phi_io the edge function of (input i, output o)
//   y_0 = sum_i phi_io(x_i),   y_{1+i} = phi_io'(x_i),   y_{1+D+i} = phi_io''(x_i)  (i < D2):
// 85 multiply-adds per 48 bytes of output instead of the 32 per input and jet-general edge of the kernels above;
// what is left is writing (forward) and reading (weight gradients) N x O x J floats once.
//   * a wave takes kSeedBatch points at a time: lane = (point, input) evaluates the 4 live bases, the activation and
//     their derivatives once and leaves them in the wave's LDS area (no block barrier in the loop);
//   * then lane = output: the records are broadcast reads, and the pair's interval - wave-uniform - names the 4 + 1
//     weights (forward: an LDS table) or running sums (weight gradients: registers, picked by a jump) it touches;
//   * weight gradients: the same walk with gY read instead of y written; lane o keeps the I x C1 sums of its
//     output in registers and adds A0 B + A1 B' + A2 B'' into the 4 + 1 of them the pair's interval names (a jump on
//     the wave-uniform interval; sums kept in an LDS table cost 2x as read-add-write and 10x as LDS float atomics);
//     the sums go out as the chunk's partial part[chunk][O][I][C1].
// ==========================================================================
constexpr int kSeedBatch = 8;                 // points per wave and batch
constexpr int kSeedRec = 16;                  // floats per evaluated (point, input): b[4], d1[4], d2[4], f0, f1, f2, c0

// lane = (point, input) of the batch at P0 -> record in stg; jets_out: where the seed jets are left, or nullptr
template <int D, int D2>
__device__ __forceinline__ void seeded_eval_batch(float* stg, const float* sK, int nk, int act, float xv, int lane,
                                                  bool eval_lane, bool live, float* jets_out) {
  constexpr int J = 1 + D + D2, I = D;
  if (!eval_lane) return;
  const int ei = lane % I;
  Basis bs;
  eval_basis_row<false>(sK + ei * (nk + 1), nk, xv, bs);
  const Act av = eval_act(act, xv);
  float4* e = reinterpret_cast<float4*>(stg + lane * kSeedRec);
  e[0] = make_float4(bs.b[0], bs.b[1], bs.b[2], bs.b[3]);
  e[1] = make_float4(bs.d1[0], bs.d1[1], bs.d1[2], bs.d1[3]);
  e[2] = make_float4(bs.d2[0], bs.d2[1], bs.d2[2], bs.d2[3]);
  e[3] = make_float4(av.f0, av.f1, av.f2, __int_as_float(bs.c0));
  if (jets_out && live) {
    float jet[J];
    jet[0] = xv;
#pragma unroll
    for (int d = 0; d < D; ++d) jet[1 + d] = (d == ei) ? 1.0f : 0.0f;
#pragma unroll
    for (int d = 0; d < D2; ++d) jet[1 + D + d] = 0.0f;
    if constexpr (J % 4 == 0) {
#pragma unroll
      for (int q = 0; q < J / 4; ++q)
        reinterpret_cast<float4*>(jets_out)[q] = make_float4(jet[4 * q], jet[4 * q + 1], jet[4 * q + 2], jet[4 * q + 3]);
    } else {
#pragma unroll
      for (int j = 0; j < J; ++j) jets_out[j] = jet[j];
    }
  }
}

#define PDEKAN_SEED_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                                     __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Block = 4 waves, each on its own stream of point batches and all the layer's outputs (64 at a time); the layer's
// transposed weights wt[I*C1][O] sit in LDS: the weights of column c0 + m are consecutive over the lanes, the
// interval is wave-uniform (scalar branch).  A point's 64 jets of a group go out through the wave's LDS tile as
// 64 * J contiguous floats: every store instruction writes 1 KiB of whole lines (a lane's own jet would be J floats
// at a stride of J).  (With the weights in registers, picked by a jump like the sums of the weight-gradient kernel,
// the kernel took 256 + 132 registers and ran at one wave per SIMD: 14 ms against 8 at the 5-asset model's size.)
constexpr int kSeedOutMax = 4;                // 64-output groups per wave

template <int D, int D2>
__global__ void __launch_bounds__(256)
fwd_seeded_kernel(const float* __restrict__ x, int64_t sn, int64_t sd, const float* __restrict__ grid,
                  const float* __restrict__ ginfo, const float* __restrict__ wt, float* __restrict__ y,
                  float* __restrict__ a_out, int64_t N, int O, int G, int act, unsigned* __restrict__ amax_out) {
  constexpr int J = 1 + D + D2, I = D;
  static_assert(kSeedBatch * I <= 64 && J % 4 == 0, "one (point, input) per lane; whole 16-byte pieces of a jet");
  const int nk = G + 2 * kOrder + 1, C = G + kOrder, C1 = C + 1;
  const int krow = (I * (nk + 1) + 3) & ~3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sW = reinterpret_cast<float*>(smem);                // [I*C1][O]
  float* sK = sW + I * C1 * O;                               // [I][nk+1]
  float* sE = sK + krow;                                     // [4 waves][kSeedBatch*I][kSeedRec]
  float* sY = sE + 4 * kSeedBatch * I * kSeedRec;            // [4 waves][64 outputs][J]: one point's jets on their way out
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  for (int t = tid; t < I * C1 * O / 4; t += 256) reinterpret_cast<float4*>(sW)[t] = reinterpret_cast<const float4*>(wt)[t];
  load_knots_to_lds(sK, grid, ginfo, 0, I, I, nk);
  __syncthreads();

  float* stg = sE + wave * (kSeedBatch * I * kSeedRec);
  float4* ytile = reinterpret_cast<float4*>(sY + wave * (64 * J));
  const int ept = lane / I, ei = lane % I, ngrp = O >> 6;
  const bool eval_lane = lane < kSeedBatch * I;
  const int64_t nbatch = (N + kSeedBatch - 1) / kSeedBatch;
  float amax[J];
#pragma unroll
  for (int j = 0; j < J; ++j) amax[j] = 0.0f;
  auto fetch_x = [&](int64_t b) {             // this lane's input value of batch b (past the end: any valid point)
    const int64_t P0 = b * kSeedBatch;
    const int64_t p = (P0 + ept < N) ? P0 + ept : N - 1;
    return eval_lane ? x[p * sn + ei * sd] : 0.0f;
  };
  const int64_t bstride = (int64_t)gridDim.x * 4;
  int64_t b = (int64_t)blockIdx.x * 4 + wave;
  float xnext = (b < nbatch) ? fetch_x(b) : 0.0f;
  for (; b < nbatch; b += bstride) {
    const int64_t P0 = b * kSeedBatch;
    const int nlive = (int)((N - P0 < kSeedBatch) ? (N - P0) : kSeedBatch);
    const float xv = xnext;
    if (b + bstride < nbatch) xnext = fetch_x(b + bstride);
    seeded_eval_batch<D, D2>(stg, sK, nk, act, xv, lane, eval_lane, ept < nlive,
                             (a_out && eval_lane) ? a_out + ((P0 + (ept < nlive ? ept : 0)) * I + ei) * J : nullptr);
    PDEKAN_SEED_WAVE_SYNC();
    for (int pt = 0; pt < nlive; ++pt) {
      float acc[kSeedOutMax][J];
#pragma unroll
      for (int g = 0; g < kSeedOutMax; ++g)
#pragma unroll
        for (int j = 0; j < J; ++j) acc[g][j] = 0.0f;
#pragma unroll
      for (int i = 0; i < I; ++i) {
        const float4* e = reinterpret_cast<const float4*>(stg + (pt * I + i) * kSeedRec);
        const float4 vb = e[0], v1 = e[1], v2 = e[2], vf = e[3];
        const int c0 = __builtin_amdgcn_readfirstlane(__float_as_int(vf.w));
        const float bb[4] = {vb.x, vb.y, vb.z, vb.w}, b1[4] = {v1.x, v1.y, v1.z, v1.w}, b2[4] = {v2.x, v2.y, v2.z, v2.w};
        const float* wrow = sW + (size_t)(i * C1) * O + lane;
#pragma unroll
        for (int g = 0; g < kSeedOutMax; ++g) {
          if (g < ngrp) {
            const float* wcol = wrow + 64 * g;
            const float wbase = wcol[(size_t)C * O];
            float p0 = wbase * vf.x, p1 = wbase * vf.y, p2 = wbase * vf.z;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int c = c0 + m;
              if (c >= 0 && c < C) {
                const float w = wcol[(size_t)c * O];
                p0 += w * bb[m]; p1 += w * b1[m]; p2 += w * b2[m];
              }
            }
            acc[g][0] += p0;
            acc[g][1 + i] = p1;
            if (i < D2) acc[g][1 + D + i] = p2;
          }
        }
      }
#pragma unroll
      for (int g = 0; g < kSeedOutMax; ++g) {
        if (g < ngrp) {
#pragma unroll
          for (int q = 0; q < J / 4; ++q)
            ytile[lane * (J / 4) + q] = make_float4(acc[g][4 * q], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]);
#pragma unroll
          for (int j = 0; j < J; ++j) absmax_into(amax[j], acc[g][j]);
          PDEKAN_SEED_WAVE_SYNC();
          float4* dst = reinterpret_cast<float4*>(y + ((P0 + pt) * O + 64 * g) * J);
#pragma unroll
          for (int q = 0; q < J / 4; ++q) dst[64 * q + lane] = ytile[64 * q + lane];
          PDEKAN_SEED_WAVE_SYNC();
        }
      }
    }
    PDEKAN_SEED_WAVE_SYNC();                   // the records are free for the next batch
  }
  if (amax_out) publish_slot_absmax<J>(amax, amax_out);
}

struct SeedRec { float4 b, d1, d2, f; };      // the record of one evaluated (point, input), as read back from LDS
__device__ __forceinline__ SeedRec read_seed_rec(const float* stg, int pair) {
  const float4* e = reinterpret_cast<const float4*>(stg + pair * kSeedRec);
  return {e[0], e[1], e[2], e[3]};
}

// the 4 live columns c0 .. c0+3 of one (point, input) into the lane's sums of that input: c0 is wave-uniform (a
// scalar register), so this is a few scalar branches and then adds into registers named at compile time.  Only the
// adds sit in the cases: the kernel below unrolls 8 points x 6 inputs of them, and with the multiply-adds inside
// its loop was 240 KB of code that ran at the speed of the instruction cache's misses.
#define PDEKAN_SEED_CASE(K)                                                                      \
  case K: {                                                                                      \
    _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                \
      if (K + m >= 0 && K + m < C) acc[(K + m) < 0 ? 0 : ((K + m) < C ? (K + m) : 0)] += v[m];   \
  } break;
template <int C>
__device__ __forceinline__ void seeded_add_columns(float (&acc)[C + 1], int c0, const float (&v)[4]) {
  static_assert(C <= 23, "cases up to grid size 20");
  switch (c0) {
    PDEKAN_SEED_CASE(-3) PDEKAN_SEED_CASE(-2) PDEKAN_SEED_CASE(-1) PDEKAN_SEED_CASE(0) PDEKAN_SEED_CASE(1)
    PDEKAN_SEED_CASE(2) PDEKAN_SEED_CASE(3) PDEKAN_SEED_CASE(4) PDEKAN_SEED_CASE(5) PDEKAN_SEED_CASE(6)
    PDEKAN_SEED_CASE(7) PDEKAN_SEED_CASE(8) PDEKAN_SEED_CASE(9) PDEKAN_SEED_CASE(10) PDEKAN_SEED_CASE(11)
    PDEKAN_SEED_CASE(12) PDEKAN_SEED_CASE(13) PDEKAN_SEED_CASE(14) PDEKAN_SEED_CASE(15) PDEKAN_SEED_CASE(16)
    PDEKAN_SEED_CASE(17) PDEKAN_SEED_CASE(18) PDEKAN_SEED_CASE(19) PDEKAN_SEED_CASE(20) PDEKAN_SEED_CASE(21)
    PDEKAN_SEED_CASE(22)
    default: break;
  }
}
#undef PDEKAN_SEED_CASE

// a: the seed jets the forward left ([N][I][J]; only the values are read), gy: [N][O][J].  Block = O / 64 waves on
// one chunk of points (blockIdx.x), wave w on outputs 64w .. 64w+63; chunk_pts is a multiple of kSeedBatch.  The gY
// jets of a point are fetched four points ahead, into the registers the point four before it has just left.
template <int D, int D2, int G>
__global__ void __launch_bounds__(256, G <= 10 ? 2 : 1)
bwd_weight_seeded_kernel(const float* __restrict__ a, const float* __restrict__ gy, const float* __restrict__ grid,
                         const float* __restrict__ ginfo, float* __restrict__ part, int64_t N, int O, int act,
                         int64_t chunk_pts) {
  constexpr int J = 1 + D + D2, I = D, Q = J / 4;
  static_assert(kSeedBatch * I <= 64 && J % 4 == 0, "one (point, input) per lane; whole 16-byte pieces of a jet");
  constexpr int nk = G + 2 * kOrder + 1, C = G + kOrder, C1 = C + 1;
  constexpr int krow = (I * (nk + 1) + 3) & ~3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sK = reinterpret_cast<float*>(smem);
  float* sE = sK + krow;                                     // [waves][kSeedBatch*I][kSeedRec]
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  load_knots_to_lds(sK, grid, ginfo, 0, I, I, nk);
  __syncthreads();

  float* stg = sE + wave * (kSeedBatch * I * kSeedRec);
  const int ept = lane / I, ei = lane % I, o = wave * 64 + lane;
  const bool eval_lane = lane < kSeedBatch * I;
  const int64_t pbeg = (int64_t)blockIdx.x * chunk_pts;
  const int64_t pend = (pbeg + chunk_pts < N) ? pbeg + chunk_pts : N;
  float acc[I][C1];                                          // sums of this lane's output: [input][column]
#pragma unroll
  for (int i = 0; i < I; ++i)
#pragma unroll
    for (int c = 0; c < C1; ++c) acc[i][c] = 0.0f;

  if (pbeg < pend) {
    auto fetch_x = [&](int64_t P0) {          // (past the end of the chunk: any valid point)
      const int64_t pe = (P0 + ept < pend) ? P0 + ept : pend - 1;
      return eval_lane ? a[(pe * I + ei) * J] : 0.0f;
    };
    auto fetch_g = [&](int64_t p, float4 (&g)[Q]) {
      const float4* src = reinterpret_cast<const float4*>(gy + (((p < pend) ? p : pend - 1) * O + o) * J);
#pragma unroll
      for (int q = 0; q < Q; ++q) g[q] = src[q];
    };
    constexpr int kRing = 4;                  // points whose gY jets are in registers or on their way
    float4 g[kRing][Q];
#pragma unroll
    for (int pt = 0; pt < kRing; ++pt) fetch_g(pbeg + pt, g[pt]);
    float xv = fetch_x(pbeg);
    for (int64_t P0 = pbeg; P0 < pend; P0 += kSeedBatch) {
      const int nlive = (int)((pend - P0 < kSeedBatch) ? (pend - P0) : kSeedBatch);
      const float xnext = fetch_x(P0 + kSeedBatch < pend ? P0 + kSeedBatch : pbeg);
      seeded_eval_batch<D, D2>(stg, sK, nk, act, xv, lane, eval_lane, false, nullptr);
      PDEKAN_SEED_WAVE_SYNC();
      for (int h = 0; h < kSeedBatch / kRing; ++h) {         // (not unrolled: the code below is large)
#pragma unroll
        for (int r = 0; r < kRing; ++r) {
          const int pt = h * kRing + r;
          if (pt < nlive) {
            float gj[J];
#pragma unroll
            for (int q = 0; q < Q; ++q) { gj[4 * q] = g[r][q].x; gj[4 * q + 1] = g[r][q].y; gj[4 * q + 2] = g[r][q].z; gj[4 * q + 3] = g[r][q].w; }
            SeedRec rec = read_seed_rec(stg, pt * I);
#pragma unroll
            for (int i = 0; i < I; ++i) {
              const float4 vb = rec.b, v1 = rec.d1, v2 = rec.d2, vf = rec.f;
              if (i + 1 < I) rec = read_seed_rec(stg, pt * I + i + 1);   // (ahead of the branches below)
              const int c0 = __builtin_amdgcn_readfirstlane(__float_as_int(vf.w));
              const float A0 = gj[0], A1 = gj[1 + i], A2 = (i < D2) ? gj[1 + D + (i < D2 ? i : 0)] : 0.0f;
              acc[i][C] += A0 * vf.x + A1 * vf.y + A2 * vf.z;
              const float v[4] = {A0 * vb.x + A1 * v1.x + A2 * v2.x, A0 * vb.y + A1 * v1.y + A2 * v2.y,
                                  A0 * vb.z + A1 * v1.z + A2 * v2.z, A0 * vb.w + A1 * v1.w + A2 * v2.w};
              seeded_add_columns<C>(acc[i], c0, v);
            }
          }
          fetch_g(P0 + pt + kRing, g[r]);                    // the same slot, four points on
        }
      }
      PDEKAN_SEED_WAVE_SYNC();
      xv = xnext;
    }
  }
  // part[chunk][o][i][c]: the words of this lane's own output
  float* dst = part + ((size_t)blockIdx.x * O + o) * I * C1;
#pragma unroll
  for (int i = 0; i < I; ++i)
#pragma unroll
    for (int c = 0; c < C1; ++c) dst[i * C1 + c] = acc[i][c];
}
#undef PDEKAN_SEED_WAVE_SYNC

// --------------------------------------------------------------------------
// sum part[chunk][total] over the chunks into out[total]: a block takes 64
// consecutive elements, its 16 waves take every 16th chunk each (coalesced 256-byte
// reads, 4 independent running sums per wave so the loads pipeline), and the
// partial sums are added in a fixed order: bitwise reproducible.
// --------------------------------------------------------------------------
constexpr int kSumWaves = 16;

__global__ void __launch_bounds__(64 * kSumWaves)
sum_partials_kernel(const float* __restrict__ part, int nchunks, size_t total, float* __restrict__ out, ChunkSelect sel) {
  __shared__ float s_r[kSumWaves][64];
  if (sel.ginfo) {                            // which run of chunks holds this call's partials: settled by the knot rows
    const bool u = all_rows_near_uniform(sel.ginfo, sel.I);
    part += (size_t)(u ? sel.first_uniform : sel.first_other) * total;
    nchunks = u ? sel.n_uniform : sel.n_other;
  }
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const size_t e = (size_t)blockIdx.x * 64 + lane;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (e < total) {
    int k = g;
    for (; k + 3 * kSumWaves < nchunks; k += 4 * kSumWaves) {
      s0 += part[(size_t)k * total + e];
      s1 += part[(size_t)(k + kSumWaves) * total + e];
      s2 += part[(size_t)(k + 2 * kSumWaves) * total + e];
      s3 += part[(size_t)(k + 3 * kSumWaves) * total + e];
    }
    for (; k < nchunks; k += kSumWaves) s0 += part[(size_t)k * total + e];
  }
  s_r[g][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && e < total) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < kSumWaves; ++w) s += s_r[w][lane];
    out[e] = s;
  }
}

// loss = inv_n * sum of the per-block partials: the first 256 threads of a block (every thread of the block must
// call), always in the same order
__device__ __forceinline__ void loss_finish_block(float (&s_w)[4], const float* __restrict__ loss_part, int nparts,
                                                  float inv_n, float* __restrict__ loss_out, int accumulate) {
  float local = 0.0f;
  if (threadIdx.x < 256) {
    for (int t = threadIdx.x; t < nparts; t += 256) local += loss_part[t];
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = local;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3])) * inv_n;
    *loss_out = accumulate ? (*loss_out + v) : v;
  }
}

// --------------------------------------------------------------------------
// The two kernels below in one launch (C + 1 <= 64): a block takes 64 / (C+1) whole edges, adds their elements over
// the chunks exactly as sum_partials_kernel does (same order, same result bit for bit) and unpacks them as
// reduce_unpack_kernel does.  One launch less per layer and step.
// --------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * kSumWaves)
sum_unpack_kernel(const float* __restrict__ part, int nchunks, ChunkSelect sel, const float* __restrict__ spline_w,
                  const float* __restrict__ scaler, float* __restrict__ g_base, float* __restrict__ g_spline,
                  float* __restrict__ g_scaler, int I, int O, int C, int accumulate, LossJob loss) {
  __shared__ float s_r[kSumWaves][64];
  __shared__ float s_sum[64];
  if (loss.out && blockIdx.x == gridDim.x - 1) {          // one block more than the edges need: the step's loss
    __shared__ float s_w[4];
    loss_finish_block(s_w, loss.part, loss.nparts, loss.inv_n, loss.out, loss.accumulate);
    return;
  }
  const int C1 = C + 1, EB = 64 / C1, edges = O * I;
  const size_t total = (size_t)edges * C1;
  if (sel.ginfo) {                            // which run of chunks holds this call's partials: settled by the knot rows
    const bool u = all_rows_near_uniform(sel.ginfo, sel.I);
    part += (size_t)(u ? sel.first_uniform : sel.first_other) * total;
    nchunks = u ? sel.n_uniform : sel.n_other;
  }
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int el = lane / C1, c = lane - el * C1;            // edge inside the block, column
  const int edge = blockIdx.x * EB + el;
  const bool live = el < EB && edge < edges;
  const size_t e = (size_t)edge * C1 + c;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (live) {
    int k = g;
    for (; k + 3 * kSumWaves < nchunks; k += 4 * kSumWaves) {
      s0 += part[(size_t)k * total + e];
      s1 += part[(size_t)(k + kSumWaves) * total + e];
      s2 += part[(size_t)(k + 2 * kSumWaves) * total + e];
      s3 += part[(size_t)(k + 3 * kSumWaves) * total + e];
    }
    for (; k < nchunks; k += kSumWaves) s0 += part[(size_t)k * total + e];
  }
  s_r[g][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g != 0) return;                          // (no barrier below: the rest is wave 0 on its own)
  float sum = 0.0f;
#pragma unroll
  for (int w = 0; w < kSumWaves; ++w) sum += s_r[w][lane];
  s_sum[lane] = sum;
  if (live) {
    if (c < C) {
      const float v = sum * (scaler ? scaler[edge] : 1.0f);
      float* dst = g_spline + (size_t)edge * C + c;
      *dst = accumulate ? (*dst + v) : v;
    } else {
      g_base[edge] = accumulate ? (g_base[edge] + sum) : sum;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (live && c == 0 && g_scaler) {
    float gs = 0.0f;
    for (int cc = 0; cc < C; ++cc) gs += s_sum[el * C1 + cc] * spline_w[(size_t)edge * C + cc];
    g_scaler[edge] = accumulate ? (g_scaler[edge] + gs) : gs;
  }
}

// --------------------------------------------------------------------------
// sum the per-chunk partials in chunk order and unpack to parameter gradients:
//   g_base[o][i]      = gW[o][i][C]
//   g_spline[o][i][c] = gW[o][i][c] * scaler[o][i]           (scaler present)
//   g_scaler[o][i]    = sum_c gW[o][i][c] * spline_w[o][i][c]
// accumulate != 0 adds to what the output buffers already hold.
// --------------------------------------------------------------------------
__global__ void reduce_unpack_kernel(const float* __restrict__ part, int nchunks,
                                     const float* __restrict__ spline_w, const float* __restrict__ scaler,
                                     float* __restrict__ g_base, float* __restrict__ g_spline,
                                     float* __restrict__ g_scaler, int I, int O, int C, int accumulate) {
  const int C1 = C + 1;
  const int edges = O * I;
  const size_t stride = (size_t)edges * C1;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < edges; e += gridDim.x * blockDim.x) {
    float gs = 0.0f;
    const float sc = scaler ? scaler[e] : 1.0f;
    for (int c = 0; c < C1; ++c) {
      float s = 0.0f;
      const float* src = part + (size_t)e * C1 + c;
      for (int k = 0; k < nchunks; ++k) s += src[k * stride];
      if (c < C) {
        const float v = s * sc;
        float* dst = g_spline + (size_t)e * C + c;
        *dst = accumulate ? (*dst + v) : v;
        gs += s * spline_w[(size_t)e * C + c];
      } else {
        g_base[e] = accumulate ? (g_base[e] + s) : s;
      }
    }
    if (g_scaler) g_scaler[e] = accumulate ? (g_scaler[e] + gs) : gs;
  }
}

// --------------------------------------------------------------------------
// Black-Scholes residual of the scalar head's jet, its squared-error loss and
// the cotangent of the head's jet (run1.py:128, :148).  Inputs x[N][d] with the
// last column = time; D = d, D2 = d-1.
//   res = V_t + r V - sum_j ( 0.5 sigma^2 x_j^2 V_jj + r x_j V_j )
//   loss_part[block] = sum (res - target)^2 ; gy = 2 (res - target) / n_global * dres/dV*
// --------------------------------------------------------------------------
template <int D, int D2>
__global__ void __launch_bounds__(256)
bs_residual_kernel(const float* __restrict__ x, int64_t stride_n, int64_t stride_d,
                   const float* __restrict__ vjet, const float* __restrict__ target,
                   float sigma, float rate, float inv_n, float* __restrict__ res_out,
                   float* __restrict__ gy, float* __restrict__ loss_part, int64_t N) {
  constexpr int J = 1 + D + D2;
  static_assert(D2 == D - 1, "assets first, time last");
  float local = 0.0f;
  const float hs2 = 0.5f * sigma * sigma;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < N;
       p += (int64_t)gridDim.x * blockDim.x) {
    float v[J];
#pragma unroll
    for (int j = 0; j < J; ++j) v[j] = vjet[p * J + j];
    float xs[D2];
#pragma unroll
    for (int d = 0; d < D2; ++d) xs[d] = x[p * stride_n + d * stride_d];
    float res = v[D] + rate * v[0];                 // V_t is direction D-1 -> slot 1+(D-1) = D
#pragma unroll
    for (int d = 0; d < D2; ++d) res = res - hs2 * xs[d] * xs[d] * v[1 + D + d] - rate * xs[d] * v[1 + d];
    const float diff = res - (target ? target[p] : 0.0f);
    local += diff * diff;
    if (res_out) res_out[p] = res;
    if (gy) {
      const float g = 2.0f * diff * inv_n;
      float* dst = gy + p * J;
      dst[0] = g * rate;
#pragma unroll
      for (int d = 0; d < D2; ++d) { dst[1 + d] = -g * rate * xs[d]; dst[1 + D + d] = -g * hs2 * xs[d] * xs[d]; }
      dst[D] = g;
    }
  }
  // block reduction in a fixed order: lanes by shuffle, waves through LDS
  __shared__ float s_w[4];
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) loss_part[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ void loss_finish_kernel(const float* __restrict__ loss_part, int nparts, float inv_n,
                                   float* __restrict__ loss_out, int accumulate) {
  __shared__ float s_w[4];
  loss_finish_block(s_w, loss_part, nparts, inv_n, loss_out, accumulate);
}

// --------------------------------------------------------------------------
// host-side launchers
// --------------------------------------------------------------------------
static inline int pow2_floor(int v) { int r = 1; while (r * 2 <= v) r *= 2; return r; }

// --------------------------------------------------------------------------
// One 4-jet view of a 12-jet tensor (6 directions, 5 second derivatives): the
// layer map is linear in the jets above the value, so a weight gradient over
// (value, 6 firsts, 5 seconds) is the sum of 5 weight gradients over
// (value, first k, [first 5 = time in pass 0], second k), k = 0..4, when the
// cotangent's value jet is taken in pass 0 only.  in[e][12] -> out[e][4].
// --------------------------------------------------------------------------
__global__ void gather_jets4_kernel(const float* __restrict__ in, float4* __restrict__ out, int64_t total, int k,
                                    int keep_value, const float* __restrict__ skip_ginfo, int skip_I) {
  if (skip_ginfo && all_rows_near_uniform(skip_ginfo, skip_I)) return;    // nobody reads the view: see JetSkip
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const float* src = in + e * 12;
    out[e] = make_float4(keep_value ? src[0] : 0.0f, src[1 + k], k == 0 ? src[6] : 0.0f, src[7 + k]);
  }
}

// the way back for input gradients: pass k of the 4-jet kernel gives (its share of d/d value, d/d first k,
// d/d time in pass 0, d/d second k); the value shares add up over the passes.  in4[e] -> out12[e][12].
__global__ void scatter_jets4_kernel(const float4* __restrict__ in, float* __restrict__ out, int64_t total, int k,
                                     const float* __restrict__ skip_ginfo, int skip_I) {
  if (skip_ginfo && all_rows_near_uniform(skip_ginfo, skip_I)) return;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = in[e];
    float* dst = out + e * 12;
    dst[0] = (k == 0) ? v.x : dst[0] + v.x;
    dst[1 + k] = v.y;
    if (k == 0) dst[6] = v.z;
    dst[7 + k] = v.w;
  }
}

void launch_scatter_jets4(const float* in4, float* out12, int64_t elems, int k, hipStream_t st, JetSkip skip) {
  int64_t blocks = (elems + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  scatter_jets4_kernel<<<(unsigned)blocks, 256, 0, st>>>(reinterpret_cast<const float4*>(in4), out12, elems, k,
                                                         skip.ginfo, skip.I);
}

void launch_gather_jets4(const float* in12, float* out4, int64_t elems, int k, bool keep_value, hipStream_t st,
                         JetSkip skip) {
  int64_t blocks = (elems + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  gather_jets4_kernel<<<(unsigned)blocks, 256, 0, st>>>(in12, reinterpret_cast<float4*>(out4), elems, k, keep_value ? 1 : 0,
                                                        skip.ginfo, skip.I);
}

// Dense bases of every feature, or their derivative of the given order with respect to x:
// out[N][I][C] = d^order B_c(x[n][i]) / dx^order (b_splines, efficient_kan/kan.py:78-111, and what autograd makes
// of it).  One thread per (point, feature): local evaluation of the 4 live bases (general knots, half-open
// intervals), written out with the zeros around them.
__global__ void basis_dense_kernel(const float* __restrict__ x, const float* __restrict__ grid, float* __restrict__ out,
                                   int64_t total, int I, int nk, int C, int order) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t % I);
    Basis bs;
    eval_basis<true>(grid + (size_t)i * nk, nk, x[t], bs);
    const float* v = order == 0 ? bs.b : (order == 1 ? bs.d1 : (order == 2 ? bs.d2 : bs.d3));
    float* dst = out + t * C;
    for (int c = 0; c < C; ++c) {
      const int m = c - bs.c0;
      dst[c] = (m >= 0 && m < 4) ? v[m] : 0.0f;
    }
  }
}

// The same for any spline order k <= kMaxAnyOrder (the layers whose order is not 3 run `forward` as its defining
// expression around this kernel): dense bases B_{c,k}, c < G + k, or their r-th derivative, on nk = G + 2k + 1 knots.
// One thread per (point, feature): the interval s with t_s <= x < t_{s+1} (half-open as kan.py:94; x outside the
// knots: all zero), then the k + 1 bases alive on it by the Cox-de Boor recursion restricted to them (kan.py:96-108)
// up to order k - r, then r steps of  D B_{i,d} = d (B_{i,d-1} / (t_{i+d} - t_i) - B_{i+1,d-1} / (t_{i+d+1} - t_{i+1})).
// w[m] stands for basis i = s - k + m; terms whose knots would fall outside the row belong to bases that do not exist.
constexpr int kMaxAnyOrder = 5;

__global__ void basis_dense_any_kernel(const float* __restrict__ x, const float* __restrict__ grid, float* __restrict__ out,
                                       int64_t total, int I, int nk, int k, int r) {
  const int C = nk - k - 1;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const float* g = grid + (size_t)(t % I) * nk;
    const float xv = x[t];
    float* dst = out + t * C;
    int s = -1;
    for (int q = 0; q + 1 < nk; ++q)
      if (xv >= g[q] && xv < g[q + 1]) s = q;
    float w[kMaxAnyOrder + 2];
#pragma unroll
    for (int m = 0; m < kMaxAnyOrder + 2; ++m) w[m] = 0.0f;
    if (s >= 0 && r <= k) {
      w[k] = 1.0f;                                         // order 0: the indicator of interval s is basis i = s
      for (int d = 1; d <= k; ++d) {
        const bool deriv = d > k - r;
#pragma unroll
        for (int m = 0; m <= kMaxAnyOrder; ++m) {
          if (m > k) continue;
          const int i = s - k + m;
          float left = 0.0f, right = 0.0f;
          if (i >= 0 && i + d < nk) {
            const float den = g[i + d] - g[i];
            left = deriv ? w[m] / den : (xv - g[i]) / den * w[m];
          }
          if (i + 1 >= 0 && i + d + 1 < nk) {
            const float den = g[i + d + 1] - g[i + 1];
            right = deriv ? -w[m + 1] / den : (g[i + d + 1] - xv) / den * w[m + 1];
          }
          w[m] = deriv ? (float)d * (left + right) : left + right;
        }
      }
    }
    for (int c = 0; c < C; ++c) {
      const int m = c - (s - k);
      float v = 0.0f;
#pragma unroll
      for (int q = 0; q <= kMaxAnyOrder; ++q) v = (q == m) ? w[q] : v;
      dst[c] = (s >= 0 && m >= 0 && m <= k) ? v : 0.0f;
    }
  }
}

void launch_basis_dense(const float* x, const float* grid, float* out, int64_t N, int I, int G, int order, hipStream_t st,
                        int spline_order) {
  const int64_t total = N * I;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (spline_order != kOrder) {
    basis_dense_any_kernel<<<(unsigned)blocks, 256, 0, st>>>(x, grid, out, total, I, G + 2 * spline_order + 1, spline_order, order);
    return;
  }
  basis_dense_kernel<<<(unsigned)blocks, 256, 0, st>>>(x, grid, out, total, I, G + 2 * kOrder + 1, G + kOrder, order);
}

// --------------------------------------------------------------------------
// curve2coeff (efficient_kan/kan.py:113-143): per input feature i the least-squares spline coefficients through the
// samples (x[p][i], y[p][i][:]), coeff[o][i][:] = argmin_c sum_p (sum_c B_c(x_pi) c - y_pio)^2, the minimal-norm one
// where the samples leave coefficients undetermined (what the reference's lstsq returns).  One block per feature:
//   - the dense bases of 256 points at a time in LDS; thread (a, b) adds their products into the Gram matrix
//     G = B^T B, thread o adds B^T y[:, i, o] into its own right-hand side - both in fp64, in a fixed order;
//   - cyclic Jacobi rotations diagonalise G (C <= 23) in LDS: G = V diag(lam) V^T;
//   - coeff = V diag(1 / lam, 0 where lam <= (rcond^2) lam_max) V^T (B^T y): the pseudo-inverse, fp64 throughout,
//     so the result carries the conditioning of B once (fp64 on the squared condition number), not of an fp32 solver.
// The samples y are either given (y != nullptr, element strides) or are the layer's own edge functions on its OLD
// grid, sum_c B_c^old(x_pi) spline_weight[o][i][c] spline_scaler[o][i] (update_grid, kan.py:173-180 and :215): then
// they are evaluated on the fly from the 4 live old bases and never stored.
// --------------------------------------------------------------------------
constexpr int kFitCMax = 24;                 // coefficients per feature: grid_size + 3 <= 23
constexpr int kFitPts = 256;

struct FitSamples {
  const float* y; int64_t sn, si, so;        // given samples, or
  const float* old_grid; const float* old_w; const float* old_scaler;   // the layer before its grid moved
};

__global__ void __launch_bounds__(256)
fit_coeff_kernel(const float* __restrict__ x, int64_t xsn, int64_t xsi, int64_t N, int I, int O, int nk,
                 const float* __restrict__ grid, FitSamples src, double rcond, float* __restrict__ coeff) {
  const int C = nk - kOrder - 1, i = blockIdx.x, tid = threadIdx.x;
  __shared__ float sK[32], sKo[32];
  __shared__ float sB[kFitPts][kFitCMax];                  // dense bases of the tile's points on the new grid
  __shared__ float sOb[kFitPts][4];                        // live bases on the old grid (0 where the column does not exist)
  __shared__ int sOc[kFitPts];                             // their first column, clamped so that c .. c+3 exist
  __shared__ float sWo[kFitCMax][256];                     // old scaled weights of this feature: [c][o]
  __shared__ double sM[kFitCMax][kFitCMax + 1], sV[kFitCMax][kFitCMax + 1], sInv[kFitCMax];
  if (tid < nk) {
    sK[tid] = grid[(size_t)i * nk + tid];
    if (!src.y) sKo[tid] = src.old_grid[(size_t)i * nk + tid];
  }
  double gm[3] = {0.0, 0.0, 0.0};
  for (int o0 = 0; o0 < O; o0 += 256) {
    const int o = o0 + tid;
    const bool o_live = o < O;
    if (!src.y) {
      const float sc = (o_live && src.old_scaler) ? src.old_scaler[(size_t)o * I + i] : 1.0f;
      for (int c = 0; c < C; ++c) sWo[c][tid] = o_live ? src.old_w[((size_t)o * I + i) * C + c] * sc : 0.0f;
    }
    double rh[kFitCMax];
#pragma unroll
    for (int a = 0; a < kFitCMax; ++a) rh[a] = 0.0;
    __syncthreads();
    for (int64_t p0 = 0; p0 < N; p0 += kFitPts) {
      {
        const int64_t p = p0 + tid;
        Basis bs = {};
        float xv = 0.0f;
        if (p < N) { xv = x[p * xsn + i * xsi]; eval_basis<false>(sK, nk, xv, bs); }
#pragma unroll
        for (int c = 0; c < kFitCMax; ++c) {
          const int m = c - bs.c0;
          sB[tid][c] = (p < N && c < C && m >= 0 && m < 4) ? bs.b[m] : 0.0f;
        }
        if (!src.y) {
          Basis bo = {};
          if (p < N) eval_basis<false>(sKo, nk, xv, bo);
          const int cc = (p < N) ? (bo.c0 < 0 ? 0 : (bo.c0 > C - 4 ? C - 4 : bo.c0)) : 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {                    // value of column cc + q
            const int m = cc + q - bo.c0;
            sOb[tid][q] = (p < N && m >= 0 && m < 4) ? bo.b[m] : 0.0f;
          }
          sOc[tid] = cc;
        }
      }
      __syncthreads();
      const int np = (N - p0 < kFitPts) ? (int)(N - p0) : kFitPts;
      if (o0 == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int e = tid + 256 * q;
          if (e < C * C) {
            const int a = e / C, b = e - a * C;
            double sum = 0.0;
            for (int p = 0; p < np; ++p) sum += (double)sB[p][a] * (double)sB[p][b];
            gm[q] += sum;
          }
        }
      }
      if (o_live) {
        for (int p = 0; p < np; ++p) {
          float yv;
          if (src.y) {
            yv = src.y[(p0 + p) * src.sn + i * src.si + o * src.so];
          } else {
            const int cc = sOc[p];
            yv = sOb[p][0] * sWo[cc][tid] + sOb[p][1] * sWo[cc + 1][tid] + sOb[p][2] * sWo[cc + 2][tid] +
                 sOb[p][3] * sWo[cc + 3][tid];
          }
          const double yd = (double)yv;
#pragma unroll
          for (int a = 0; a < kFitCMax; ++a) rh[a] += (double)sB[p][a] * yd;
        }
      }
      __syncthreads();
    }
    if (o0 == 0) {
      // G -> sM, V = identity, then cyclic Jacobi sweeps (fixed count: quadratic convergence, C <= 23)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int e = tid + 256 * q;
        if (e < C * C) { const int a = e / C, b = e - a * C; sM[a][b] = gm[q]; sV[a][b] = (a == b) ? 1.0 : 0.0; }
      }
      __syncthreads();
      for (int sweep = 0; sweep < 12; ++sweep)
        for (int pp = 0; pp + 1 < C; ++pp)
          for (int qq = pp + 1; qq < C; ++qq) {
            const double app = sM[pp][pp], aqq = sM[qq][qq], apq = sM[pp][qq];
            double cs = 1.0, sn = 0.0, tt = 0.0;
            if (fabs(apq) > 1e-300) {
              const double tau = (aqq - app) / (2.0 * apq);
              tt = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
              cs = 1.0 / sqrt(1.0 + tt * tt);
              sn = tt * cs;
            }
            __syncthreads();
            if (tid < C) {
              const int k = tid;
              const double vkp = sV[k][pp], vkq = sV[k][qq];
              sV[k][pp] = cs * vkp - sn * vkq;
              sV[k][qq] = sn * vkp + cs * vkq;
              if (k != pp && k != qq) {
                const double akp = sM[k][pp], akq = sM[k][qq];
                const double n1 = cs * akp - sn * akq, n2 = sn * akp + cs * akq;
                sM[k][pp] = n1; sM[pp][k] = n1; sM[k][qq] = n2; sM[qq][k] = n2;
              } else if (k == pp) {
                sM[pp][pp] = app - tt * apq; sM[pp][qq] = 0.0; sM[qq][pp] = 0.0;
              } else {
                sM[qq][qq] = aqq + tt * apq;
              }
            }
            __syncthreads();
          }
      double lmax = 0.0;
      for (int k = 0; k < C; ++k) lmax = fmax(lmax, sM[k][k]);
      if (tid < C) sInv[tid] = (sM[tid][tid] > rcond * rcond * lmax && sM[tid][tid] > 0.0) ? 1.0 / sM[tid][tid] : 0.0;
      __syncthreads();
    }
    if (o_live) {
      double tk[kFitCMax];
#pragma unroll
      for (int k = 0; k < kFitCMax; ++k) {
        double sum = 0.0;
        if (k < C) {
#pragma unroll
          for (int a = 0; a < kFitCMax; ++a) sum += (a < C) ? sV[a][k] * rh[a] : 0.0;
          sum *= sInv[k];
        }
        tk[k] = sum;
      }
      float* dst = coeff + ((size_t)o * I + i) * C;
      for (int c = 0; c < C; ++c) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < kFitCMax; ++k) sum += (k < C) ? sV[c][k] * tk[k] : 0.0;
        dst[c] = (float)sum;
      }
    }
    __syncthreads();
  }
}

bool launch_fit_coefficients(const float* x, int64_t xsn, int64_t xsi, int64_t N, int I, int O, int G, const float* grid,
                             const float* y, int64_t ysn, int64_t ysi, int64_t yso, const float* old_grid,
                             const float* old_w, const float* old_scaler, float* coeff, hipStream_t st) {
  if (G + kOrder > kFitCMax - 1) return false;
  FitSamples src = {y, ysn, ysi, yso, old_grid, old_w, old_scaler};
  const int C = G + kOrder;
  const double rcond = 1.1920928955078125e-07 * (double)(N > C ? N : C);   // lstsq's default for fp32: eps * max(m, n)
  fit_coeff_kernel<<<I, 256, 0, st>>>(x, xsn, xsi, N, I, O, G + 2 * kOrder + 1, grid, src, rcond, coeff);
  return true;
}

void launch_grid_info(const LayerView& L, float* ginfo, hipStream_t st) {
  grid_info_kernel<<<(L.I + 63) / 64, 64, 0, st>>>(L.grid, ginfo, L.I, L.nk());
}

// ginfo_out (or nullptr): also classify the layer's knot rows into it (launch_grid_info in the same launch)
void launch_pack_weights(const LayerView& L, float* wp, float* wt, hipStream_t st, float* ginfo_out) {
  const int total = L.O * L.I * (L.C() + 1);
  const int blocks = (total + 255) / 256;
  pack_weights_kernel<<<blocks < 1024 ? blocks : 1024, 256, 0, st>>>(L.base_w, L.spline_w, L.scaler, wp, wt,
                                                                  L.I, L.O, L.C(), L.grid, ginfo_out, L.nk());
}

// wp / wt / ginfo: one pointer per layer.  Returns false (nothing launched) when the model has more layers than one
// launch takes; the caller then packs layer by layer.
bool launch_pack_model(const LayerView* layers, int n_layers, float* const* wp, float* const* wt, float* const* ginfo,
                       unsigned* clear, int clear_words, hipStream_t st) {
  if (n_layers > kPackLayersMax) return false;
  PackModelArgs args;
  int blocks = 1;
  for (int l = 0; l < n_layers; ++l) {
    const LayerView& L = layers[l];
    args.layer[l] = PackLayer{L.base_w, L.spline_w, L.scaler, L.grid, wp[l], wt[l], ginfo[l], L.I, L.O, L.C(), L.nk()};
    const int b = (L.O * L.I * (L.C() + 1) + 255) / 256;
    if (b > blocks) blocks = b;
  }
  for (int l = n_layers; l < kPackLayersMax; ++l) args.layer[l] = args.layer[0];
  args.clear = clear; args.clear_words = clear_words;
  if (blocks > 1024) blocks = 1024;
  pack_model_kernel<<<dim3(blocks, n_layers), 256, 0, st>>>(args);
  return true;
}

template <int D, int D2>
static void seed_t(const float* x, int64_t sn, int64_t sd, float* a, int64_t N, int I, hipStream_t st) {
  const int64_t total = N * I;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  seed_jets_kernel<D, D2><<<(int)blocks, 256, 0, st>>>(x, sn, sd, a, N, I);
}

// input features per launch of the streaming forward / input-gradient kernels: bounds their LDS footprint
// (the forward launcher narrows it further when a large grid would push a window past 60 KB)
constexpr int kGenericWindow = 256;

// input features per launch of the streaming forward
template <int J>
static int fwd_window(const LayerView& L) {
  int IB = L.I < kGenericWindow ? L.I : kGenericWindow;
  const int per_feature = (int)sizeof(EdgeRec) + J * 4 + (L.nk() + 1) * 4;
  if (IB * per_feature > 60 * 1024) IB = (60 * 1024) / per_feature;    // large grids: narrower window
  return IB;
}

template <int D, int D2>
static void fwd_t(const LayerView& L, const float* a, const float* wt, float* y, int64_t N, hipStream_t st, bool skip,
                  unsigned* amax_out, SeedSrc seed) {
  constexpr int J = 1 + D + D2;
  const int IB = fwd_window<J>(L);
  int PT = 256 / IB; if (PT < 1) PT = 1; if (PT > 64) PT = 64;
  int NIC = 1;
  if (PT * L.O < 256) {
    NIC = pow2_floor(256 / (PT * L.O));
    const int cap = pow2_floor(IB);
    if (NIC > cap) NIC = cap;
    if (NIC > 64) NIC = 64;
  }
  const size_t lds = (size_t)PT * IB * (sizeof(EdgeRec) + J * 4) + (size_t)IB * (L.nk() + 1) * 4;
  const int64_t blocks = (N + PT - 1) / PT;
  for (int i0 = 0; i0 < L.I; i0 += IB)
    fwd_generic_kernel<D, D2><<<(unsigned)blocks, 256, lds, st>>>(a, L.grid, L.ginfo, wt, y, N, L.I, L.O, L.G, L.act,
                                                                  PT, NIC, i0, IB, i0 > 0, skip ? 1 : 0,
                                                                  (i0 + IB >= L.I) ? amax_out : nullptr, seed);
}

template <int D, int D2>
static void bwd_input_t(const LayerView& L, const float* a, const float* gy, const float* wp, float* ga,
                        int64_t N, hipStream_t st, bool skip) {
  const int IB = L.I < kGenericWindow ? L.I : kGenericWindow;
  const int64_t total = N * IB;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  const size_t lds = (size_t)IB * (L.nk() + 1) * 4;
  for (int i0 = 0; i0 < L.I; i0 += IB)
    bwd_input_generic_kernel<D, D2><<<(int)blocks, 256, lds, st>>>(a, gy, L.grid, L.ginfo, wp, ga, N, L.I, L.O, L.G,
                                                                   L.act, i0, IB, skip ? 1 : 0);
}

static inline int pow2_ceil(int v) { int r = 1; while (r < v) r *= 2; return r; }

static BwTile bw_tile(int I, int O) {
  BwTile T;
  T.TI = pow2_ceil(I); if (T.TI > 16) T.TI = 16;
  T.TO = pow2_ceil(O); if (T.TO > 256 / T.TI) T.TO = 256 / T.TI;
  T.PG = 256 / (T.TI * T.TO);
  T.PT = 256 / T.TI; if (T.PT > 128) T.PT = 128;
  if (T.PT < T.PG) T.PT = T.PG;
  return T;
}

int generic_weight_chunks(int64_t N, int I, int O) {
  // enough blocks to fill the chip, few enough that the partials stay small
  const BwTile T = bw_tile(I, O);
  const int tiles = ((I + T.TI - 1) / T.TI) * ((O + T.TO - 1) / T.TO);
  int64_t want = (2048 + tiles - 1) / tiles;
  const int64_t max_by_pts = (N + 127) / 128;            // (small batches are latency-bound: short chunks finish sooner)
  if (want > max_by_pts) want = max_by_pts;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

template <int D, int D2>
static void bwd_weight_t(const LayerView& L, const float* a, const float* gy, float* part, int nchunks,
                         int64_t N, hipStream_t st, bool skip) {
  constexpr int J = 1 + D + D2;
  const BwTile T = bw_tile(L.I, L.O);
  int64_t chunk = (N + nchunks - 1) / nchunks;
  chunk = (chunk + T.PT - 1) / T.PT * T.PT;
  dim3 g((L.I + T.TI - 1) / T.TI, (L.O + T.TO - 1) / T.TO, nchunks);
  const size_t lds = sizeof(EdgeRec) * T.PT * T.TI + (size_t)T.PT * T.TI * J * 4 + (size_t)T.TI * (L.nk() + 1) * 4 +
                     (size_t)L.C1() * 256 * 4;
  bwd_weight_generic_kernel<D, D2><<<g, 256, lds, st>>>(a, gy, L.grid, L.ginfo, part, N, L.I, L.O, L.G, L.act, chunk,
                                                        T, skip ? 1 : 0);
}

template <int D, int D2>
static void residual_t(const float* x, int64_t sn, int64_t sd, const float* vjet, const float* target,
                       float sigma, float rate, float inv_n, float* res, float* gy, float* loss_part,
                       int nblocks, int64_t N, hipStream_t st) {
  bs_residual_kernel<D, D2><<<nblocks, 256, 0, st>>>(x, sn, sd, vjet, target, sigma, rate, inv_n, res, gy,
                                                     loss_part, N);
}

// first layers the seeded kernels take (fwd_seeded_kernel, bwd_weight_seeded_kernel): as many inputs as directions,
// so that the seeds make input i direction i; 64, 128 or 256 outputs (the groups of 64 divide a block's 4 waves)
static size_t fwd_seeded_lds_bytes(const LayerView& L) {
  const int krow = (L.I * (L.nk() + 1) + 3) & ~3;
  return ((size_t)L.I * L.C1() * L.O + krow + 4 * ((size_t)kSeedBatch * L.I * kSeedRec + 64 * 12)) * 4;
}

bool seeded_first_layer(int D, int D2, const LayerView& L) {
  if (D != 6 || D2 != 5 || L.I != D || L.O % 64 != 0 || L.O > 256 || L.O == 192) return false;
  if (L.G != 5 && L.G != 10 && L.G != 20) return false;      // (the kernels are built per grid size)
  const char* off = getenv("PDEKAN_DISABLE_SEEDED_FIRST");   // A/B switch for tests and profiling
  if (off && off[0] == '1') return false;
  return fwd_seeded_lds_bytes(L) <= 128 * 1024;
}

static void fwd_seeded_launch(const LayerView& L, const float* wt, float* y, int64_t N, hipStream_t st,
                              unsigned* amax_out, const SeedSrc& seed) {
  auto k = fwd_seeded_kernel<6, 5>;
  const size_t lds = fwd_seeded_lds_bytes(L);
  ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
  const int64_t nbatch = (N + kSeedBatch - 1) / kSeedBatch;
  int64_t blocks = (nbatch + 3) / 4;
  const int64_t resident = 256 * (lds <= 72 * 1024 ? 2 : 1);    // blocks the chip holds at once
  if (blocks > resident) blocks = resident;
  k<<<(unsigned)blocks, 256, lds, st>>>(seed.x, seed.sn, seed.sd, L.grid, L.ginfo, wt, y, seed.a_out, N, L.O, L.G, L.act,
                                        amax_out);
}

template <int G>
static void bwd_weight_seeded_launch_g(const LayerView& L, const float* a, const float* gy, float* part, int nchunks,
                                       int64_t N, hipStream_t st) {
  const int waves = L.O / 64;
  const int krow = (L.I * (L.nk() + 1) + 3) & ~3;
  const size_t lds = ((size_t)krow + (size_t)waves * kSeedBatch * L.I * kSeedRec) * 4;
  int64_t chunk = (N + nchunks - 1) / nchunks;
  chunk = (chunk + kSeedBatch - 1) / kSeedBatch * kSeedBatch;
  bwd_weight_seeded_kernel<6, 5, G><<<(unsigned)nchunks, 64 * waves, lds, st>>>(a, gy, L.grid, L.ginfo, part, N, L.O, L.act,
                                                                               chunk);
}

static void bwd_weight_seeded_launch(const LayerView& L, const float* a, const float* gy, float* part, int nchunks,
                                     int64_t N, hipStream_t st) {
  if (L.G == 5) bwd_weight_seeded_launch_g<5>(L, a, gy, part, nchunks, N, st);
  else if (L.G == 10) bwd_weight_seeded_launch_g<10>(L, a, gy, part, nchunks, N, st);
  else bwd_weight_seeded_launch_g<20>(L, a, gy, part, nchunks, N, st);
}

#define PDEKAN_DISPATCH_JET(D_, D2_, CALL)                       \
  if ((D_) == 0 && (D2_) == 0) { CALL(0, 0); }                   \
  else if ((D_) == 2 && (D2_) == 1) { CALL(2, 1); }              \
  else if ((D_) == 6 && (D2_) == 5) { CALL(6, 5); }              \
  else return false;

bool launch_seed_jets(int D, int D2, const float* x, int64_t sn, int64_t sd, float* a, int64_t N, int I,
                      hipStream_t st) {
#define CALL(A, B) seed_t<A, B>(x, sn, sd, a, N, I, st)
  PDEKAN_DISPATCH_JET(D, D2, CALL)
#undef CALL
  return true;
}

// seed (x != nullptr): only for layers fwd_generic_takes_seed() accepts (one window of input features)
bool launch_fwd_generic(int D, int D2, const LayerView& L, const float* a, const float* wt, float* y, int64_t N,
                        hipStream_t st, bool skip_near_uniform, unsigned* amax_out, SeedSrc seed) {
  if (seed.x && !skip_near_uniform && seeded_first_layer(D, D2, L)) {
    fwd_seeded_launch(L, wt, y, N, st, amax_out, seed);
    return true;
  }
#define CALL(A, B) fwd_t<A, B>(L, a, wt, y, N, st, skip_near_uniform, amax_out, seed)
  PDEKAN_DISPATCH_JET(D, D2, CALL)
#undef CALL
  return true;
}

bool fwd_generic_takes_seed(int D, int D2, const LayerView& L) {
#define CALL(A, B) return fwd_window<1 + A + B>(L) >= L.I
  PDEKAN_DISPATCH_JET(D, D2, CALL)
#undef CALL
  return false;
}

bool launch_bwd_input_generic(int D, int D2, const LayerView& L, const float* a, const float* gy, const float* wp,
                              float* ga, int64_t N, hipStream_t st, bool skip_near_uniform) {
#define CALL(A, B) bwd_input_t<A, B>(L, a, gy, wp, ga, N, st, skip_near_uniform)
  PDEKAN_DISPATCH_JET(D, D2, CALL)
#undef CALL
  return true;
}

bool launch_bwd_weight_generic(int D, int D2, const LayerView& L, const float* a, const float* gy, float* part,
                               int nchunks, int64_t N, hipStream_t st, bool skip_near_uniform, bool seeded_input) {
  if (seeded_input && !skip_near_uniform && seeded_first_layer(D, D2, L)) {
    bwd_weight_seeded_launch(L, a, gy, part, nchunks, N, st);
    return true;
  }
#define CALL(A, B) bwd_weight_t<A, B>(L, a, gy, part, nchunks, N, st, skip_near_uniform)
  PDEKAN_DISPATCH_JET(D, D2, CALL)
#undef CALL
  return true;
}

// loss (out != nullptr): the loss reduction of launch_loss_finish rides in the same launch where it can
void launch_reduce_unpack(const LayerView& L, const float* part, int nchunks, float* sum_buf, float* g_base,
                          float* g_spline, float* g_scaler, int accumulate, hipStream_t st, ChunkSelect sel,
                          LossJob loss) {
  const int edges = L.O * L.I;
  const float* src = part;
  if ((nchunks > 1 || sel.ginfo) && L.C() + 1 <= 64) {   // both steps in one launch
    const int EB = 64 / (L.C() + 1);
    sum_unpack_kernel<<<(unsigned)((edges + EB - 1) / EB) + (loss.out ? 1 : 0), 64 * kSumWaves, 0, st>>>(
        part, nchunks, sel, L.spline_w, L.scaler, g_base, g_spline, g_scaler, L.I, L.O, L.C(), accumulate, loss);
    return;
  }
  if (loss.out) launch_loss_finish(loss.part, loss.nparts, loss.inv_n, loss.out, loss.accumulate, st);
  if (nchunks > 1 || sel.ginfo) {
    const size_t total = L.packed_floats();
    sum_partials_kernel<<<(unsigned)((total + 63) / 64), 64 * kSumWaves, 0, st>>>(part, nchunks, total, sum_buf, sel);
    src = sum_buf;
  }
  const int blocks = (edges + 127) / 128;
  reduce_unpack_kernel<<<blocks, 128, 0, st>>>(src, 1, L.spline_w, L.scaler, g_base, g_spline, g_scaler,
                                               L.I, L.O, L.C(), accumulate);
}

bool launch_bs_residual(int D, const float* x, int64_t sn, int64_t sd, const float* vjet, const float* target,
                        float sigma, float rate, float inv_n, float* res, float* gy, float* loss_part,
                        int nblocks, int64_t N, hipStream_t st) {
  if (D == 2) residual_t<2, 1>(x, sn, sd, vjet, target, sigma, rate, inv_n, res, gy, loss_part, nblocks, N, st);
  else if (D == 6) residual_t<6, 5>(x, sn, sd, vjet, target, sigma, rate, inv_n, res, gy, loss_part, nblocks, N, st);
  else return false;
  return true;
}

void launch_loss_finish(const float* loss_part, int nparts, float inv_n, float* loss_out, int accumulate,
                        hipStream_t st) {
  loss_finish_kernel<<<1, 256, 0, st>>>(loss_part, nparts, inv_n, loss_out, accumulate);
}

}  // namespace pdekan
