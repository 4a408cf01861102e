// One training step of the physics-informed DeepONet the reference scripts build around four small KANs, in four
// launches (run1.py:75-149 the class, :217-233 the loop body, :478-482 the networks): the layer-by-layer path takes
// about a hundred launches for the same step, and at the scripts' batch size of 100 (run1.py:470-473) a step is the
// sum of its launch latencies.
//
//   s(x, t)   = sum_k B_k T_k(x, t)                    run1.py:109-117   T = model5(x, t), B = model4(row v)
//   v_r       = model1(u1_r) * model2(u2_r)            run1.py:92-98     one value per sensor row r
//   bc loss   = mean((out - s)^2)                      run1.py:134-141
//   pde loss  = mean((out - res)^2), res = s_t - sigma^2/2 x^2 s_xx - r x s_x + r s      run1.py:120-129, :145-149
//   loss      = pde loss + bc_weight * bc loss         run1.py:224, then loss.backward() (:225)
//
// The reference feeds model4 n copies of the row v and differentiates s with jacrev / hessian; B does not depend on
// (x, t), so model4 runs on the one row and the derivatives of s are B times the input jets of the trunk.
//
//   don_sensor_forward_kernel   EP lanes per sensor row (one per edge): model1 and model2 forward, y1, y2, v to memory,
//                               and what v[r], as input feature r of model4's first layer, adds to that layer's outputs
//   don_trunk_kernel            block = a share of the boundary and collocation points.  Every block first adds up
//                               the outputs of model4's first layer (the features' shares, in a fixed order) and
//                               runs model4's upper layers on wave 0 (B: the same bits in every block), then the
//                               trunk on its points with 4-jets (lane = (point, edge)), residual, squared errors, the
//                               trunk's backward into a per-lane gradient table in LDS; one partial per block: the
//                               table's sums, the cotangent of B, the two sums of squares
//   don_sensor_backward_kernel  every block first adds the trunk blocks' cotangents of B and takes them back through
//                               model4's upper layers on wave 0 (block 0 writes those layers' gradients); then per
//                               sensor row: gradients of model4's first layer for that input feature (each
//                               feature's weights belong to one row: no sums), the cotangent of v, model1 and
//                               model2 forward again and backward into per-lane tables; one partial per block
//   don_finish_kernel           adds the blocks' partials in a fixed order: model1 / model2 / trunk gradients
//                               (unpacked), the three losses
// Every sum has a fixed order: results are bitwise reproducible.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <initializer_list>
#include "kan_math.h"
#include "kan_launch.h"

namespace pdekan {

constexpr int kDonWidth = 8;                 // widest layer boundary of a network held in LDS
constexpr int kDonBlocksMax = 1024;
constexpr int kDonSensorThreads = 256;
constexpr int kDonTrunkThreads = 256;
constexpr int kDonTrunkBlocksMax = 64;

struct DonLayer {
  int I, O, G, act;
  const float *grid, *base_w, *spline_w, *scaler;
  float *g_base, *g_spline, *g_scaler;
  int wofs;                                  // first packed weight of the layer: wp[o][i][C1], C1 = G + 4
  int kofs;                                  // first word of its knot rows in LDS: [i][nk + 1] (the last word: 1/h or 0)
  int aofs;                                  // first activation word of its INPUT boundary: [feature][jet]
  int tofs;                                  // first of its gradient-table rows: [c][lanes]
};
struct DonNet {
  int nl, P, kwords, slots, trows, edges;    // packed weights, knot words, activation words per row, table rows, edges
  DonLayer L[kDonLayersMax];
};

#define DON_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__device__ __forceinline__ int round4(int v) { return (v + 3) & ~3; }

// packed weights (spline_weight * spline_scaler, base weight behind them) and knot rows of a network into LDS, by the
// nthreads threads tid = 0 .. of the block; a block barrier, then don_stage_spacing, then another barrier must follow
__device__ __forceinline__ void don_stage(const DonNet& M, float* sW, float* sK, int tid, int nthreads) {
  for (int l = 0; l < M.nl; ++l) {
    const DonLayer& L = M.L[l];
    const int C = L.G + 3, C1 = C + 1, nk = L.G + 7;
    for (int k = tid; k < L.O * L.I * C1; k += nthreads) {
      const int edge = k / C1, c = k - edge * C1;
      sW[L.wofs + k] = (c < C) ? L.spline_w[(size_t)edge * C + c] * (L.scaler ? L.scaler[edge] : 1.0f) : L.base_w[edge];
    }
    for (int k = tid; k < L.I * (nk + 1); k += nthreads) {
      const int i = k / (nk + 1), kk = k - i * (nk + 1);
      if (kk < nk) sK[L.kofs + k] = L.grid[(size_t)i * nk + kk];
    }
  }
}
// (thread 8 l + i takes input i of layer l: 32 threads per network, so callers hand each network its own 32)
__device__ __forceinline__ void don_stage_spacing(const DonNet& M, float* sK, int tid) {
  const int i = tid & 7;
#pragma unroll
  for (int l = 0; l < kDonLayersMax; ++l) {
    if (l >= M.nl || (tid >> 3) != l) continue;
    const DonLayer& L = M.L[l];
    const int nk = L.G + 7;
    if (i < L.I) sK[L.kofs + i * (nk + 1) + nk] = equal_spacing_inv_h(sK + L.kofs + i * (nk + 1), nk);
  }
}

// Forward of a network on the row whose lanes e = 0 .. EP-1 (one wave, or a part of one) call this together: the
// input jets [feature][4] stand at myA + L[0].aofs; every boundary's jets are left in myA for the backward pass.
__device__ __forceinline__ void don_forward(const DonNet& M, const float* sW, const float* sK, float* myA, float* myC,
                                            int e) {
#pragma unroll
  for (int l = 0; l < kDonLayersMax; ++l) {
    if (l >= M.nl) break;
    const DonLayer& L = M.L[l];
    const int C = L.G + 3, C1 = C + 1, nk = L.G + 7;
    const bool edge_live = e < L.O * L.I;
    const int o = edge_live ? e / L.I : 0, i = edge_live ? e - o * L.I : 0;
    {
      const float4 a = *reinterpret_cast<const float4*>(myA + L.aofs + i * 4);
      Basis bs;
      eval_basis_row<false>(sK + L.kofs + i * (nk + 1), nk, a.x, bs);
      const Act av = eval_act(L.act, a.x);
      const float* w = sW + L.wofs + (o * L.I + i) * C1;
      const float wb = w[C];
      float e0 = av.f0 * wb, e1 = av.f1 * wb, e2 = av.f2 * wb;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int c = bs.c0 + m;
        const float wv = (c >= 0 && c < C) ? w[c] : 0.0f;
        e0 += wv * bs.b[m]; e1 += wv * bs.d1[m]; e2 += wv * bs.d2[m];
      }
      *reinterpret_cast<float4*>(myC + e * 4) = make_float4(e0, e1 * a.y, e1 * a.z, e2 * a.y * a.y + e1 * a.w);
    }
    DON_WAVE_SYNC();
    if (e < L.O) {                            // lane e adds the I edges of output e, in order
      float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < L.I; ++k) {
        const float4 c = *reinterpret_cast<const float4*>(myC + (e * L.I + k) * 4);
        y.x += c.x; y.y += c.y; y.z += c.z; y.w += c.w;
      }
      *reinterpret_cast<float4*>(myA + L.aofs + L.I * 4 + e * 4) = y;
    }
    DON_WAVE_SYNC();
  }
}

// Backward of the same: on entry myG [output][4] holds the cotangent of the network's output jets, the lane adds its
// edge's parameter gradients into its own column myT[row * TS] of the gradient table; with input_cot the cotangent of
// the network's INPUT jets is left in myG [input][4].
template <int TS>
__device__ __forceinline__ void don_backward(const DonNet& M, const float* sW, const float* sK, const float* myA,
                                             float* myG, float* myC, float* myT, int e, bool input_cot) {
#pragma unroll
  for (int lr = 0; lr < kDonLayersMax; ++lr) {
    const int l = kDonLayersMax - 1 - lr;
    if (l >= M.nl) continue;
    const DonLayer& L = M.L[l];
    const int C = L.G + 3, C1 = C + 1, nk = L.G + 7;
    const bool edge_live = e < L.O * L.I;
    const int o = edge_live ? e / L.I : 0, i = edge_live ? e - o * L.I : 0;
    {
      const float4 a = *reinterpret_cast<const float4*>(myA + L.aofs + i * 4);
      const float4 g = *reinterpret_cast<const float4*>(myG + o * 4);
      Basis bs;
      eval_basis_row<true>(sK + L.kofs + i * (nk + 1), nk, a.x, bs);
      const Act av = eval_act(L.act, a.x);
      const float q = a.y * a.y;
      const float* w = sW + L.wofs + (o * L.I + i) * C1;
      // <gy, dy> = A0 phi + A1 phi' + A2 phi''
      const float A0 = g.x, A1 = g.y * a.y + g.z * a.z + g.w * a.w, A2 = g.w * q;
      float* t = myT + (size_t)L.tofs * TS;
      const float wb = w[C];
      float e1 = av.f1 * wb, e2 = av.f2 * wb, e3 = av.f3 * wb;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int c = bs.c0 + m;
        const bool in = edge_live && c >= 0 && c < C;
        const int cc = in ? c : 0;
        const float wv = in ? w[cc] : 0.0f;
        e1 += wv * bs.d1[m]; e2 += wv * bs.d2[m]; e3 += wv * bs.d3[m];
        if (in) t[cc * TS] += A0 * bs.b[m] + A1 * bs.d1[m] + A2 * bs.d2[m];
      }
      if (edge_live) t[C * TS] += A0 * av.f0 + A1 * av.f1 + A2 * av.f2;
      // y0 = phi, y1 = phi' a1, y2 = phi' a2, y3 = phi'' a1^2 + phi' a3
      *reinterpret_cast<float4*>(myC + e * 4) =
          make_float4(g.x * e1 + A1 * e2 + A2 * e3, g.y * e1 + 2.0f * g.w * a.y * e2, g.z * e1, g.w * e1);
    }
    if (l == 0 && !input_cot) continue;
    DON_WAVE_SYNC();
    if (e < L.I) {                            // lane e adds the O edges that start at input e, in order
      float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < L.O; ++k) {
        const float4 c = *reinterpret_cast<const float4*>(myC + (k * L.I + e) * 4);
        ga.x += c.x; ga.y += c.y; ga.z += c.z; ga.w += c.w;
      }
      *reinterpret_cast<float4*>(myG + e * 4) = ga;
    }
    DON_WAVE_SYNC();
  }
}

// which layer packed word k of a network belongs to
__device__ __forceinline__ int don_layer_of(const DonNet& M, int k) {
  int l = 0;
  while (l + 1 < M.nl && k >= M.L[l + 1].wofs) ++l;
  return l;
}

// gradients of a network from its summed gradient table (LDS, word (row, edge) at table[row * TS + edge]):
// spline_weight takes the scaler, spline_scaler the sum over the columns of gW * spline_weight (kan.py:145-151
// differentiated); by nthreads threads; the table must be complete
template <int TS>
__device__ __forceinline__ void don_unpack(const DonNet& M, const float* table, int tid, int nthreads) {
  for (int k = tid; k < M.P; k += nthreads) {
    const DonLayer& L = M.L[don_layer_of(M, k)];
    const int C = L.G + 3, C1 = C + 1;
    const int edge = (k - L.wofs) / C1, c = (k - L.wofs) - edge * C1;
    const float sum = table[(size_t)(L.tofs + c) * TS + edge];
    if (c < C) L.g_spline[(size_t)edge * C + c] = sum * (L.scaler ? L.scaler[edge] : 1.0f);
    else L.g_base[edge] = sum;
  }
  for (int l = 0; l < M.nl; ++l) {
    const DonLayer& L = M.L[l];
    if (!L.scaler || !L.g_scaler) continue;
    const int C = L.G + 3;
    for (int edge = tid; edge < L.O * L.I; edge += nthreads) {
      float gs = 0.0f;
      for (int c = 0; c < C; ++c) gs += table[(size_t)(L.tofs + c) * TS + edge] * L.spline_w[(size_t)edge * C + c];
      L.g_scaler[edge] = gs;
    }
  }
}

// ---- sensors, forward: y1[r] = model1(u1[r]), y2[r] = model2(u2[r]), v[r] = y1[r] * y2[r]
template <int EP>
__global__ void __launch_bounds__(kDonSensorThreads)
don_sensor_forward_kernel(DonLayer F, DonNet M1, DonNet M2, const float* __restrict__ u1, const float* __restrict__ u2,
                          int m, float* __restrict__ y1, float* __restrict__ y2, float* __restrict__ v,
                          float* __restrict__ first_out) {
  constexpr int TB = kDonSensorThreads, PB = TB / EP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW1 = smem;
  float* sK1 = sW1 + round4(M1.P);
  float* sW2 = sK1 + round4(M1.kwords);
  float* sK2 = sW2 + round4(M2.P);
  float* sA = sK2 + round4(M2.kwords);
  const int slots = M1.slots > M2.slots ? M1.slots : M2.slots;
  float* sC = sA + PB * slots;
  const int tid = threadIdx.x, ps = tid / EP, e = tid % EP;
  don_stage(M1, sW1, sK1, tid, TB);
  don_stage(M2, sW2, sK2, tid, TB);
  __syncthreads();
  don_stage_spacing(M1, sK1, tid);
  don_stage_spacing(M2, sK2, tid - 32);
  __syncthreads();
  float* myA = sA + ps * slots;
  float* myC = sC + (tid - e) * 4;
  for (int r0 = blockIdx.x * PB; r0 < m; r0 += gridDim.x * PB) {
    const int r = r0 + ps;
    const bool live = r < m;
    const int rr = live ? r : m - 1;
    if (e < M1.L[0].I) *reinterpret_cast<float4*>(myA + e * 4) = make_float4(u1[(size_t)rr * M1.L[0].I + e], 0.f, 0.f, 0.f);
    DON_WAVE_SYNC();
    don_forward(M1, sW1, sK1, myA, myC, e);
    const float a1 = myA[M1.slots - 4];
    DON_WAVE_SYNC();
    if (e < M2.L[0].I) *reinterpret_cast<float4*>(myA + e * 4) = make_float4(u2[(size_t)rr * M2.L[0].I + e], 0.f, 0.f, 0.f);
    DON_WAVE_SYNC();
    don_forward(M2, sW2, sK2, myA, myC, e);
    const float a2 = myA[M2.slots - 4];
    DON_WAVE_SYNC();
    if (e == 0 && live) { y1[r] = a1; y2[r] = a2; v[r] = a1 * a2; }
    // what input feature r of model4's first layer adds to each of its outputs: first_out[o][r]
    if (e < F.O && live) {
      const int C = F.G + 3, nk = F.G + 7;
      const float x = a1 * a2;
      Basis bs;
      eval_basis<false>(F.grid + (size_t)r * nk, nk, x, bs);
      const Act av = eval_act(F.act, x);
      const size_t edge = (size_t)e * m + r;
      const float sc = F.scaler ? F.scaler[edge] : 1.0f;
      const float* w = F.spline_w + edge * C;
      float s = av.f0 * F.base_w[edge];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = bs.c0 + k;
        if (c >= 0 && c < C) s += (w[c] * sc) * bs.b[k];
      }
      first_out[edge] = s;
    }
  }
}

// ---- the trunk: block = a share of the points
template <int EP>
__global__ void __launch_bounds__(kDonTrunkThreads)
don_trunk_kernel(DonLayer F, DonNet M4, DonNet M5, const float* __restrict__ first_out, int m,
                 const float* __restrict__ x_bc, const float* __restrict__ t_bc, const float* __restrict__ out_bc, int n_bc,
                 const float* __restrict__ x_pde, const float* __restrict__ t_pde, const float* __restrict__ out_pde,
                 int n_pde, float sigma, float rate, float bc_weight, float* __restrict__ y4_out,
                 float* __restrict__ part5, float* __restrict__ part_b, float* __restrict__ part_l,
                 float* __restrict__ s_pred, float* __restrict__ res_out, int want_grads) {
  constexpr int TB = kDonTrunkThreads, PB = TB / EP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW5 = smem;
  float* sK5 = sW5 + round4(M5.P);
  float* sW4 = sK5 + round4(M5.kwords);
  float* sK4 = sW4 + round4(M4.P);
  float* sA = sK4 + round4(M4.kwords);       // [PB][slots of the trunk]
  float* sG = sA + PB * M5.slots;            // [PB][8][4]
  float* sC = sG + PB * kDonWidth * 4;       // [TB][4]
  float* sA4 = sC + TB * 4;                  // wave 0's row through model4's upper layers
  float* sC4 = sA4 + round4(M4.slots > 0 ? M4.slots : 4);   // [64][4]
  float* sT = sC4 + 64 * 4;                  // [max(trows of the trunk, 8)][TB]
  __shared__ float sY4[kDonWidth], sB[kDonWidth];
  __shared__ float sRed2[kDonWidth * 64];
  __shared__ float s_loss[TB / 64][2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int ps = tid / EP, e = tid % EP;
  don_stage(M5, sW5, sK5, tid, TB);
  don_stage(M4, sW4, sK4, tid, TB);
  // ---- first layer of model4 on the row v: the features' shares of its O outputs, added in a fixed order
  {
    float acc[kDonWidth];
#pragma unroll
    for (int o = 0; o < kDonWidth; ++o) acc[o] = 0.0f;
    for (int i = tid; i < m; i += TB) {
#pragma unroll
      for (int o = 0; o < kDonWidth; ++o)
        if (o < F.O) acc[o] += first_out[(size_t)o * m + i];
    }
#pragma unroll
    for (int o = 0; o < kDonWidth; ++o)
      if (o < F.O) sT[o * TB + tid] = acc[o];
  }
  __syncthreads();
  don_stage_spacing(M5, sK5, tid);
  don_stage_spacing(M4, sK4, tid - 32);
  for (int k = tid; k < F.O * 64; k += TB) {
    const int o = k >> 6, j = k & 63;
    float s = 0.0f;
    for (int q = 0; q < TB / 64; ++q) s += sT[o * TB + j * (TB / 64) + q];
    sRed2[k] = s;
  }
  __syncthreads();
  if (tid < F.O) {
    float s = 0.0f;
    for (int j = 0; j < 64; ++j) s += sRed2[tid * 64 + j];
    sY4[tid] = s;
    if (blockIdx.x == 0) y4_out[tid] = s;
  }
  {
    const int rows = M5.trows > kDonWidth ? M5.trows : kDonWidth;
    for (int k = tid; k < rows * TB; k += TB) sT[k] = 0.0f;
  }
  __syncthreads();
  // ---- upper layers of model4 on wave 0: B
  const int n_out = M5.L[M5.nl - 1].O;
  if (tid < 64) {
    if (M4.nl == 0) {
      if (tid < n_out) sB[tid] = sY4[tid];
    } else {
      if (tid < F.O) *reinterpret_cast<float4*>(sA4 + tid * 4) = make_float4(sY4[tid], 0.f, 0.f, 0.f);
      DON_WAVE_SYNC();
      don_forward(M4, sW4, sK4, sA4, sC4, tid);
      if (tid < n_out) sB[tid] = sA4[M4.slots - 4 * n_out + 4 * tid];
    }
  }
  __syncthreads();
  // ---- the trunk on this block's share of the boundary points and the collocation points
  float* myA = sA + ps * M5.slots;
  float* myG = sG + ps * kDonWidth * 4;
  float* myC = sC + (tid - e) * 4;
  float* myT = sT + tid;
  const float* outJ = myA + M5.slots - 4 * n_out;
  const float half_s2 = 0.5f * sigma * sigma;
  const float inv_bc = 1.0f / (float)n_bc, inv_pde = 1.0f / (float)n_pde;
  float loss_bc = 0.0f, loss_pde = 0.0f, gB = 0.0f;
  const int n_all = n_bc + n_pde;
  for (int q0 = blockIdx.x * PB; q0 < n_all; q0 += gridDim.x * PB) {
    const int q = q0 + ps;
    const bool live = q < n_all;
    const int qq = live ? q : n_all - 1;
    const bool is_bc = qq < n_bc;
    const int idx = is_bc ? qq : qq - n_bc;
    const float S = is_bc ? x_bc[idx] : x_pde[idx], tt = is_bc ? t_bc[idx] : t_pde[idx];
    if (e < 2) *reinterpret_cast<float4*>(myA + e * 4) = (e == 0) ? make_float4(S, 1.0f, 0.0f, 0.0f) : make_float4(tt, 0.0f, 1.0f, 0.0f);
    DON_WAVE_SYNC();
    don_forward(M5, sW5, sK5, myA, myC, e);
    if (e == 0) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < n_out; ++k) {
        const float b = sB[k];
        const float4 T = *reinterpret_cast<const float4*>(outJ + 4 * k);
        s.x += b * T.x; s.y += b * T.y; s.z += b * T.z; s.w += b * T.w;
      }
      float4 g;
      if (is_bc) {
        const float diff = s.x - out_bc[idx];
        if (live) {
          if (s_pred) s_pred[idx] = s.x;
          loss_bc += diff * diff;
        }
        g = make_float4(live ? 2.0f * bc_weight * inv_bc * diff : 0.0f, 0.f, 0.f, 0.f);
      } else {
        const float cS = -rate * S, cSS = -half_s2 * S * S;
        const float res = s.z + cSS * s.w + cS * s.y + rate * s.x;
        const float diff = res - out_pde[idx];
        if (live) {
          if (res_out) res_out[idx] = res;
          loss_pde += diff * diff;
        }
        const float gg = live ? 2.0f * inv_pde * diff : 0.0f;
        g = make_float4(gg * rate, gg * cS, gg, gg * cSS);
      }
      *reinterpret_cast<float4*>(myC) = g;
    }
    if (!want_grads) continue;
    DON_WAVE_SYNC();
    if (e < n_out) {
      const float4 g = *reinterpret_cast<const float4*>(myC);
      const float4 T = *reinterpret_cast<const float4*>(outJ + 4 * e);
      gB += (g.x * T.x + g.y * T.y) + (g.z * T.z + g.w * T.w);
      const float b = sB[e];
      *reinterpret_cast<float4*>(myG + e * 4) = make_float4(b * g.x, b * g.y, b * g.z, b * g.w);
    }
    DON_WAVE_SYNC();
    don_backward<TB>(M5, sW5, sK5, myA, myG, myC, myT, e, false);
    DON_WAVE_SYNC();
  }
  // ---- the block's partial: sums of squares, cotangent of B (the points' shares in order), the table's sums
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    loss_bc += __shfl_xor(loss_bc, d);
    loss_pde += __shfl_xor(loss_pde, d);
  }
  if (lane == 0) { s_loss[tid >> 6][0] = loss_bc; s_loss[tid >> 6][1] = loss_pde; }
  __syncthreads();                            // (every wave is done with the contribution words and its table column)
  if (tid == 0) {
    float b = 0.0f, p = 0.0f;
    for (int k = 0; k < TB / 64; ++k) { b += s_loss[k][0]; p += s_loss[k][1]; }
    part_l[blockIdx.x * 2 + 0] = b;
    part_l[blockIdx.x * 2 + 1] = p;
  }
  if (!want_grads) return;
  sC[tid] = (e < n_out) ? gB : 0.0f;
  __syncthreads();
  if (tid < n_out) {
    float s = 0.0f;
    for (int k = 0; k < PB; ++k) s += sC[k * EP + tid];
    part_b[blockIdx.x * kDonWidth + tid] = s;
  }
  float* dst = part5 + (size_t)blockIdx.x * M5.P;
  for (int k = tid; k < M5.P; k += TB) {
    const DonLayer& L = M5.L[don_layer_of(M5, k)];
    const int C1 = L.G + 4;
    const int edge = (k - L.wofs) / C1, c = (k - L.wofs) - edge * C1;
    const float* row = sT + (size_t)(L.tofs + c) * TB + edge;
    float sum = 0.0f;
#pragma unroll
    for (int q = 0; q < PB; ++q) sum += row[q * EP];
    dst[k] = sum;
  }
}

// ---- sensors, backward
template <int EP>
__global__ void __launch_bounds__(kDonSensorThreads)
don_sensor_backward_kernel(DonLayer F, DonNet M1, DonNet M2, DonNet M4, const float* __restrict__ u1,
                           const float* __restrict__ u2, int m, const float* __restrict__ y1,
                           const float* __restrict__ y2, const float* __restrict__ v, const float* __restrict__ y4,
                           const float* __restrict__ part_b, int trunk_blocks, int n_out, float* __restrict__ part) {
  constexpr int TB = kDonSensorThreads, PB = TB / EP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW1 = smem;
  float* sK1 = sW1 + round4(M1.P);
  float* sW2 = sK1 + round4(M1.kwords);
  float* sK2 = sW2 + round4(M2.P);
  float* sW4 = sK2 + round4(M2.kwords);
  float* sK4 = sW4 + round4(M4.P);
  float* sA = sK4 + round4(M4.kwords);
  const int slots = M1.slots > M2.slots ? M1.slots : M2.slots;
  float* sG = sA + PB * slots;
  float* sC = sG + PB * kDonWidth * 4;
  float* sA4 = sC + TB * 4;                  // wave 0's row through model4's upper layers
  float* sG4 = sA4 + round4(M4.slots > 0 ? M4.slots : 4);
  float* sC4 = sG4 + kDonWidth * 4;          // [64][4]
  float* sT4 = sC4 + 64 * 4;                 // [trows of model4's upper layers][64]
  float* sT1 = sT4 + M4.trows * 64;          // [trows of model1][TB], then [trows of model2][TB]
  float* sT2 = sT1 + M1.trows * TB;
  __shared__ float sGY[kDonWidth];           // cotangent of the outputs of model4's first layer
  const int tid = threadIdx.x, ps = tid / EP, e = tid % EP;
  don_stage(M1, sW1, sK1, tid, TB);
  don_stage(M2, sW2, sK2, tid, TB);
  don_stage(M4, sW4, sK4, tid, TB);
  for (int k = tid; k < M4.trows * 64 + (M1.trows + M2.trows) * TB; k += TB) sT4[k] = 0.0f;
  __syncthreads();
  don_stage_spacing(M1, sK1, tid);
  don_stage_spacing(M2, sK2, tid - 32);
  don_stage_spacing(M4, sK4, tid - 64);
  __syncthreads();
  // ---- cotangent of B (the trunk blocks' shares, in order) back through model4's upper layers, on wave 0
  if (tid < 64) {
    float gb = 0.0f;
    if (tid < n_out)
      for (int b = 0; b < trunk_blocks; ++b) gb += part_b[b * kDonWidth + tid];
    if (M4.nl == 0) {
      if (tid < n_out) sGY[tid] = gb;
    } else {
      if (tid < F.O) *reinterpret_cast<float4*>(sA4 + tid * 4) = make_float4(y4[tid], 0.f, 0.f, 0.f);
      DON_WAVE_SYNC();
      don_forward(M4, sW4, sK4, sA4, sC4, tid);
      if (tid < n_out) *reinterpret_cast<float4*>(sG4 + tid * 4) = make_float4(gb, 0.f, 0.f, 0.f);
      DON_WAVE_SYNC();
      don_backward<64>(M4, sW4, sK4, sA4, sG4, sC4, sT4 + tid, tid, true);
      if (tid < F.O) sGY[tid] = sG4[tid * 4];
      if (blockIdx.x == 0) {                  // (one row: a lane's column is the sum)
        DON_WAVE_SYNC();
        don_unpack<64>(M4, sT4, tid, 64);
      }
    }
  }
  __syncthreads();
  float* myA = sA + ps * slots;
  float* myG = sG + ps * kDonWidth * 4;
  float* myC = sC + (tid - e) * 4;
  const int C = F.G + 3, nk = F.G + 7;
  for (int r0 = blockIdx.x * PB; r0 < m; r0 += gridDim.x * PB) {
    const int r = r0 + ps;
    const bool live = r < m;
    const int rr = live ? r : m - 1;
    // first layer of model4, input feature r: its gradients, and d(loss)/d v[r]
    if (e < F.O) {
      const float x = v[rr];
      Basis bs;
      eval_basis<false>(F.grid + (size_t)rr * nk, nk, x, bs);
      const Act av = eval_act(F.act, x);
      const size_t edge = (size_t)e * m + rr;
      const float sc = F.scaler ? F.scaler[edge] : 1.0f;
      const float* w = F.spline_w + edge * C;
      const float gy = sGY[e];
      float gs = 0.0f, dphi = av.f1 * F.base_w[edge];
      if (live)                               // columns whose basis is zero at x
        for (int c = 0; c < C; ++c)
          if (c < bs.c0 || c > bs.c0 + 3) F.g_spline[edge * C + c] = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = bs.c0 + k;
        if (c >= 0 && c < C) {
          const float wc = w[c];
          if (live) F.g_spline[edge * C + c] = (gy * bs.b[k]) * sc;
          gs += (gy * bs.b[k]) * wc;
          dphi += (wc * sc) * bs.d1[k];
        }
      }
      if (live) {
        F.g_base[edge] = gy * av.f0;
        if (F.scaler && F.g_scaler) F.g_scaler[edge] = gs;
      }
      myC[e * 4] = gy * dphi;
    }
    DON_WAVE_SYNC();
    float gv = 0.0f;
    for (int o = 0; o < F.O; ++o) gv += myC[o * 4];
    const float c1 = live ? gv * y2[rr] : 0.0f, c2 = live ? gv * y1[rr] : 0.0f;
    DON_WAVE_SYNC();
    if (e < M1.L[0].I) *reinterpret_cast<float4*>(myA + e * 4) = make_float4(u1[(size_t)rr * M1.L[0].I + e], 0.f, 0.f, 0.f);
    if (e == 0) *reinterpret_cast<float4*>(myG) = make_float4(c1, 0.f, 0.f, 0.f);
    DON_WAVE_SYNC();
    don_forward(M1, sW1, sK1, myA, myC, e);
    don_backward<TB>(M1, sW1, sK1, myA, myG, myC, sT1 + tid, e, false);
    DON_WAVE_SYNC();
    if (e < M2.L[0].I) *reinterpret_cast<float4*>(myA + e * 4) = make_float4(u2[(size_t)rr * M2.L[0].I + e], 0.f, 0.f, 0.f);
    if (e == 0) *reinterpret_cast<float4*>(myG) = make_float4(c2, 0.f, 0.f, 0.f);
    DON_WAVE_SYNC();
    don_forward(M2, sW2, sK2, myA, myC, e);
    don_backward<TB>(M2, sW2, sK2, myA, myG, myC, sT2 + tid, e, false);
    DON_WAVE_SYNC();
  }
  __syncthreads();
  float* dst = part + (size_t)blockIdx.x * (M1.P + M2.P);
  auto sums = [&](const DonNet& M, const float* table, float* out) {
    for (int k = tid; k < M.P; k += TB) {
      const DonLayer& L = M.L[don_layer_of(M, k)];
      const int C1 = L.G + 4;
      const int edge = (k - L.wofs) / C1, c = (k - L.wofs) - edge * C1;
      const float* row = table + (size_t)(L.tofs + c) * TB + edge;
      float sum = 0.0f;
#pragma unroll
      for (int q = 0; q < PB; ++q) sum += row[q * EP];
      out[k] = sum;
    }
  };
  sums(M1, sT1, dst);
  sums(M2, sT2, dst + M1.P);
}

// block = one edge of model1, model2 or the trunk, thread = (column c = tid & 31, slice s = tid >> 5): slice s adds the
// partials of blocks s, s + 8, ..., column c then adds its 8 slices in order, and the edge is unpacked; the last
// block adds the trunk blocks' sums of squares into the three losses.  first_block: the block index of block 0
// (loss only: the last block alone)
__global__ void __launch_bounds__(256)
don_finish_kernel(DonNet M1, DonNet M2, DonNet M5, const float* __restrict__ part, int sensor_blocks,
                  const float* __restrict__ part5, const float* __restrict__ part_l, int trunk_blocks, float inv_bc,
                  float inv_pde, float bc_weight, float* __restrict__ losses, int first_block) {
  __shared__ float s_sl[8][33];
  __shared__ float s_col[32];
  const int tid = threadIdx.x, c = tid & 31, sl = tid >> 5;
  const int blk = blockIdx.x + first_block;
  auto edge_block = [&](const DonNet& M, int edge, const float* src0, size_t stride, int nblocks) {
    int l = 0;
    while (l + 1 < M.nl && edge >= M.L[l].O * M.L[l].I) { edge -= M.L[l].O * M.L[l].I; ++l; }
    const DonLayer& L = M.L[l];
    const int C = L.G + 3, C1 = C + 1;
    const bool col = c < C1;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};     // (four loads in flight: every one is a trip to memory)
    if (col) {
      const float* src = src0 + L.wofs + edge * C1 + c;
      for (int b0 = sl; b0 < nblocks; b0 += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int b = b0 + 8 * u;
          if (b < nblocks) v[u] += src[(size_t)b * stride];
        }
      }
    }
    s_sl[sl][c] = (v[0] + v[1]) + (v[2] + v[3]);
    __syncthreads();
    if (tid < 32) {
      float gw = 0.0f;
      for (int k = 0; k < 8; ++k) gw += s_sl[k][tid];
      s_col[tid] = (col && c < C) ? gw * L.spline_w[(size_t)edge * C + c] : 0.0f;
      if (col) {
        if (c < C) L.g_spline[(size_t)edge * C + c] = gw * (L.scaler ? L.scaler[edge] : 1.0f);
        else L.g_base[edge] = gw;
      }
    }
    __syncthreads();
    if (tid == 0 && L.scaler && L.g_scaler) {
      float gs = 0.0f;
      for (int k = 0; k < C; ++k) gs += s_col[k];
      L.g_scaler[edge] = gs;
    }
  };
  const size_t stride_s = (size_t)M1.P + M2.P;
  if (blk < M1.edges) edge_block(M1, blk, part, stride_s, sensor_blocks);
  else if (blk < M1.edges + M2.edges) edge_block(M2, blk - M1.edges, part + M1.P, stride_s, sensor_blocks);
  else if (blk < M1.edges + M2.edges + M5.edges) edge_block(M5, blk - M1.edges - M2.edges, part5, (size_t)M5.P, trunk_blocks);
  else if (tid == 0 && losses) {
    float b = 0.0f, p = 0.0f;
    for (int k = 0; k < trunk_blocks; ++k) { b += part_l[k * 2]; p += part_l[k * 2 + 1]; }
    b *= inv_bc; p *= inv_pde;
    losses[0] = p + bc_weight * b;
    losses[1] = p;
    losses[2] = b;
  }
}

// ---- host side
static bool don_disabled() {                 // A/B switch for tests and profiling
  const char* v = getenv("PDEKAN_DISABLE_DEEPONET_STEP");
  return v && v[0] == '1';
}

static DonLayer don_layer(const LayerView& L, const SmallGrads* g) {
  DonLayer S;
  S.I = L.I; S.O = L.O; S.G = L.G; S.act = L.act;
  S.grid = L.grid; S.base_w = L.base_w; S.spline_w = L.spline_w; S.scaler = L.scaler;
  S.g_base = g ? g->base_w : nullptr;
  S.g_spline = g ? g->spline_w : nullptr;
  S.g_scaler = g ? g->scaler : nullptr;
  S.wofs = S.kofs = S.aofs = S.tofs = 0;
  return S;
}

// layers [first, nl) of a network as one LDS-resident net
static DonNet don_net(const DonNetView& V, int first) {
  DonNet M;
  M.nl = V.nl - first;
  int P = 0, kw = 0, slots = 0, trows = 0, edges = 0;
  for (int l = first; l < V.nl; ++l) {
    const LayerView& L = V.layers[l];
    DonLayer& S = M.L[l - first];
    S = don_layer(L, V.grads ? V.grads + l : nullptr);
    S.wofs = P; S.kofs = kw; S.aofs = slots; S.tofs = trows;
    P += (int)L.packed_floats();
    kw += L.I * (L.nk() + 1);
    slots += L.I * 4;
    trows += L.C1();
    edges += L.O * L.I;
  }
  if (M.nl > 0) slots += V.layers[V.nl - 1].O * 4;
  M.P = P; M.kwords = kw; M.slots = slots; M.trows = trows; M.edges = edges;
  return M;
}

static int don_max_edges(const DonNet& M) {
  int edges = 0;
  for (int l = 0; l < M.nl; ++l) edges = M.L[l].O * M.L[l].I > edges ? M.L[l].O * M.L[l].I : edges;
  return edges;
}
static int r4(int v) { return (v + 3) & ~3; }

struct DonPlan {
  DonLayer F;                                // first layer of model4
  DonNet M1, M2, M4, M5;                     // model1, model2, upper layers of model4, trunk
  int ep_s, ep_t, blocks_s, blocks_t;
  size_t lds_fwd, lds_bwd, lds_trunk;
};

static const char* don_plan(const DonNetView nets[4], int64_t m, int64_t n_bc, int64_t n_pde, DonPlan& Z) {
  if (don_disabled()) return "disabled by PDEKAN_DISABLE_DEEPONET_STEP";
  for (int k = 0; k < 4; ++k) {
    if (nets[k].nl < 1 || nets[k].nl > kDonLayersMax) return "a network has no layers or more than 4";
    for (int l = (k == 2 ? 1 : 0); l < nets[k].nl; ++l) {
      const LayerView& L = nets[k].layers[l];
      if (L.I > kDonWidth || L.O > kDonWidth || L.G > 20) return "a layer is wider than 8 or its grid finer than 20";
    }
    for (int l = 0; l + 1 < nets[k].nl; ++l)
      if (nets[k].layers[l].O != nets[k].layers[l + 1].I) return "layer widths of a network do not chain";
    if ((nets[k].grads != nullptr) != (nets[0].grads != nullptr)) return "gradients for all four networks or for none";
  }
  const LayerView& F = nets[2].layers[0];
  if (m < 1 || m > (1 << 20) || F.I != m) return "model4 takes one input per sensor row (at most 2^20)";
  if (F.O > kDonWidth || F.G > 20) return "model4's first layer has more than 8 outputs or a grid finer than 20";
  if (nets[0].layers[nets[0].nl - 1].O != 1 || nets[1].layers[nets[1].nl - 1].O != 1)
    return "model1 and model2 must have one output";
  if (nets[3].layers[0].I != 2) return "the trunk takes (x, t)";
  if (nets[2].layers[nets[2].nl - 1].O != nets[3].layers[nets[3].nl - 1].O)
    return "model4 and model5 must have the same output width";
  if (n_bc < 1 || n_pde < 1 || n_bc + n_pde > (1 << 16)) return "1 .. 65536 points in all";
  Z.F = don_layer(F, nets[2].grads ? nets[2].grads : nullptr);
  Z.M1 = don_net(nets[0], 0);
  Z.M2 = don_net(nets[1], 0);
  Z.M4 = don_net(nets[2], 1);
  Z.M5 = don_net(nets[3], 0);
  const int es = don_max_edges(Z.M1) > don_max_edges(Z.M2) ? don_max_edges(Z.M1) : don_max_edges(Z.M2);
  Z.ep_s = es <= 16 ? 16 : 64;
  Z.ep_t = don_max_edges(Z.M5) <= 16 ? 16 : 64;
  const int pbs = kDonSensorThreads / Z.ep_s;
  const int slots_s = Z.M1.slots > Z.M2.slots ? Z.M1.slots : Z.M2.slots;
  const size_t stage_s = (size_t)r4(Z.M1.P) + r4(Z.M1.kwords) + r4(Z.M2.P) + r4(Z.M2.kwords);
  const size_t stage_4 = (size_t)r4(Z.M4.P) + r4(Z.M4.kwords);
  const size_t row_4 = (size_t)r4(Z.M4.slots > 0 ? Z.M4.slots : 4);
  Z.lds_fwd = (stage_s + (size_t)pbs * slots_s + (size_t)kDonSensorThreads * 4) * sizeof(float);
  Z.lds_bwd = (stage_s + stage_4 + (size_t)pbs * slots_s + (size_t)pbs * kDonWidth * 4 + (size_t)kDonSensorThreads * 4 +
               row_4 + kDonWidth * 4 + 64 * 4 + (size_t)Z.M4.trows * 64 +
               (size_t)(Z.M1.trows + Z.M2.trows) * kDonSensorThreads) * sizeof(float);
  if (Z.lds_bwd > 150 * 1024) return "model1 / model2 / model4 tables exceed the LDS of a CU";
  int64_t b = (m + pbs - 1) / pbs;
  Z.blocks_s = (int)(b > kDonBlocksMax ? kDonBlocksMax : b);
  const int pbt = kDonTrunkThreads / Z.ep_t;
  const int rows = Z.M5.trows > kDonWidth ? Z.M5.trows : kDonWidth;
  Z.lds_trunk = ((size_t)r4(Z.M5.P) + r4(Z.M5.kwords) + stage_4 + (size_t)pbt * Z.M5.slots + (size_t)pbt * kDonWidth * 4 +
                 (size_t)kDonTrunkThreads * 4 + row_4 + 64 * 4 + (size_t)rows * kDonTrunkThreads) * sizeof(float);
  if (Z.lds_trunk > 150 * 1024) return "the trunk's tables exceed the LDS of a CU";
  b = (n_bc + n_pde + pbt - 1) / pbt;
  Z.blocks_t = (int)(b > kDonTrunkBlocksMax ? kDonTrunkBlocksMax : b);
  return nullptr;
}

const char* deeponet_step_refusal(const DonNetView nets[4], int64_t m, int64_t n_bc, int64_t n_pde) {
  DonPlan Z;
  return don_plan(nets, m, n_bc, n_pde, Z);
}

// y1, y2, v [m], the outputs of model4's first layer, the features' shares of those outputs [O][m], then the partials
// of the sensor blocks and of the trunk blocks (gradient sums, cotangent of B, sums of squares); each on its own 256 bytes
static size_t don_align(size_t floats) { return (floats + 63) / 64 * 64; }
size_t deeponet_step_ws_floats(const DonNetView nets[4], int64_t m, int64_t n_bc, int64_t n_pde) {
  DonPlan Z;
  if (don_plan(nets, m, n_bc, n_pde, Z)) return 0;
  return 3 * don_align((size_t)m) + don_align(kDonWidth) + don_align((size_t)Z.F.O * m) +
         don_align((size_t)Z.blocks_s * (Z.M1.P + Z.M2.P)) + don_align((size_t)Z.blocks_t * Z.M5.P) +
         don_align((size_t)Z.blocks_t * kDonWidth) + don_align((size_t)Z.blocks_t * 2);
}

bool launch_deeponet_step(const DonNetView nets[4], const float* u1, const float* u2, int64_t m, const float* x_bc,
                          const float* t_bc, const float* out_bc, int64_t n_bc, const float* x_pde, const float* t_pde,
                          const float* out_pde, int64_t n_pde, float sigma, float rate, float bc_weight, float* losses,
                          float* s_pred, float* res_out, float* ws, hipStream_t st) {
  DonPlan Z;
  if (don_plan(nets, m, n_bc, n_pde, Z)) return false;
  float* y1 = ws;
  float* y2 = y1 + don_align((size_t)m);
  float* v = y2 + don_align((size_t)m);
  float* y4 = v + don_align((size_t)m);
  float* first_out = y4 + don_align(kDonWidth);
  float* part = first_out + don_align((size_t)Z.F.O * m);
  float* part5 = part + don_align((size_t)Z.blocks_s * (Z.M1.P + Z.M2.P));
  float* part_b = part5 + don_align((size_t)Z.blocks_t * Z.M5.P);
  float* part_l = part_b + don_align((size_t)Z.blocks_t * kDonWidth);
  const int want = nets[0].grads ? 1 : 0;
  const int n_out = Z.M5.L[Z.M5.nl - 1].O;
  auto sensors_fwd = [&](auto kern) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), Z.lds_fwd);
    kern<<<Z.blocks_s, kDonSensorThreads, Z.lds_fwd, st>>>(Z.F, Z.M1, Z.M2, u1, u2, (int)m, y1, y2, v, first_out);
  };
  if (Z.ep_s == 16) sensors_fwd(don_sensor_forward_kernel<16>);
  else sensors_fwd(don_sensor_forward_kernel<64>);
  auto trunk = [&](auto kern) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), Z.lds_trunk);
    kern<<<Z.blocks_t, kDonTrunkThreads, Z.lds_trunk, st>>>(Z.F, Z.M4, Z.M5, first_out, (int)m, x_bc, t_bc, out_bc, (int)n_bc,
                                                             x_pde, t_pde, out_pde, (int)n_pde, sigma, rate, bc_weight, y4,
                                                             part5, part_b, part_l, s_pred, res_out, want);
  };
  if (Z.ep_t == 16) trunk(don_trunk_kernel<16>);
  else trunk(don_trunk_kernel<64>);
  const int edges = Z.M1.edges + Z.M2.edges + Z.M5.edges;
  const float inv_bc = 1.0f / (float)n_bc, inv_pde = 1.0f / (float)n_pde;
  if (want) {
    auto sensors_bwd = [&](auto kern) {
      ensure_dynamic_lds(reinterpret_cast<const void*>(kern), Z.lds_bwd);
      kern<<<Z.blocks_s, kDonSensorThreads, Z.lds_bwd, st>>>(Z.F, Z.M1, Z.M2, Z.M4, u1, u2, (int)m, y1, y2, v, y4, part_b,
                                                              Z.blocks_t, n_out, part);
    };
    if (Z.ep_s == 16) sensors_bwd(don_sensor_backward_kernel<16>);
    else sensors_bwd(don_sensor_backward_kernel<64>);
  }
  don_finish_kernel<<<want ? edges + 1 : 1, 256, 0, st>>>(Z.M1, Z.M2, Z.M5, part, Z.blocks_s, part5, part_l, Z.blocks_t,
                                                           inv_bc, inv_pde, bc_weight, losses, want ? 0 : edges);
  return true;
}

}  // namespace pdekan
