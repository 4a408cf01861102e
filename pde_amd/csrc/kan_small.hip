// One Black-Scholes training step of a SMALL KAN in two launches: models like the reference scripts' own
// (KAN([2, 5, 1]), the trunk KAN([2, 6, 1]) of run1.py:478-482; batch 100, run1.py:470-473), where a step of the
// layer-by-layer path is the sum of its launch latencies.
//
// Reference computation: KAN.forward (efficient_kan/kan.py:274-279, layers kan.py:153-166), its derivatives with
// respect to (S, t) (run1.py:121-124), the residual (run1.py:128), the squared-error loss (run1.py:148) and
// loss.backward() (run1.py:225).
//
//   small_step_kernel    lane = (point, edge): EP = 16 or 64 lanes per point, one per edge (o, i) of the layer at hand,
//                        so the serial work per point is one basis evaluation and one edge per layer and direction.
//                        The packed weights and knot rows of every layer sit in LDS.  Forward: a lane evaluates its
//                        edge's contribution to the 4 jets of output o, the I lanes of an output are added (in order)
//                        through LDS.  Backward: the lane adds its edge's parameter gradients into its OWN column of a
//                        gradient table in LDS (no two lanes touch a word: no atomics) and the O lanes of an input are
//                        added into that input's cotangent.  At the end the block adds the columns of the lanes that
//                        hold the same edge in a fixed order: one partial gradient vector (+ loss share) per block.
//   small_finish_kernel  adds the blocks' partials in a fixed order, unpacks them to the parameters' gradients
//                        (spline_weight: times the scaler; spline_scaler: sum over the columns of gW * spline_weight;
//                        kan.py:145-151 differentiated) and writes the loss.
// Every sum has a fixed order: results are bitwise reproducible.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kan_math.h"
#include "kan_launch.h"

namespace pdekan {

constexpr int kSmallThreadsMax = 1024;      // threads per block: 1024, or 256 where the gradient table of 1024 lanes does not fit
constexpr int kSmallWidth = 8;               // widest layer boundary
constexpr int kSmallBlocksMax = 2048;

struct SmallLayer {
  int I, O, G, act;
  const float *grid, *base_w, *spline_w, *scaler;
  float *g_base, *g_spline, *g_scaler;
  int wofs;                                  // first packed weight of the layer: wp[o][i][C1], C1 = G + 4
  int kofs;                                  // first word of its knot rows in LDS: [i][nk + 1] (the last word: 1/h or 0)
  int aofs;                                  // first activation word of its INPUT boundary: [feature][jet]
  int tofs;                                  // first row of its gradient-table rows: [c][TB lanes]
};
struct SmallModel {
  int nl, P, kwords, slots, trows;           // packed weights, knot words, activation words per point, table rows
  SmallLayer L[kSmallLayersMax];
};

// EP lanes per point: a wave holds 64 / EP points; TB threads per block
template <int EP, int TB>
__global__ void __launch_bounds__(TB)
small_step_kernel(SmallModel M, const float* __restrict__ x, int64_t sn, int64_t sd, const float* __restrict__ target,
                  int64_t N, float sigma, float rate, float inv_n, float* __restrict__ res_out,
                  float* __restrict__ part, int want_grads) {
  constexpr int kSmallThreads = TB;
  constexpr int PB = TB / EP;                // points per block and pass
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;                          // [P]
  float* sK = sW + ((M.P + 3) & ~3);         // knot rows
  float* sA = sK + ((M.kwords + 3) & ~3);    // [PB][slots]: activations of every layer boundary of a point (16-byte rows)
  float* sG = sA + PB * M.slots;             // [PB][8][4]: cotangents of the boundary the backward pass stands at
  float* sC = sG + PB * kSmallWidth * 4;     // [TB][4]: what each lane's edge contributes
  float* sT = sC + kSmallThreads * 4;        // [trows][TB]: each lane's parameter gradients
  const int tid = threadIdx.x, lane = tid & 63;
  const int ps = tid / EP, e = tid % EP;     // point slot in the block, edge index in the layer
  for (int l = 0; l < M.nl; ++l) {
    const SmallLayer& L = M.L[l];
    const int C = L.G + 3, C1 = C + 1, nk = L.G + 7;
    for (int k = tid; k < L.O * L.I * C1; k += kSmallThreads) {
      const int edge = k / C1, c = k - edge * C1;
      sW[L.wofs + k] = (c < C) ? L.spline_w[(size_t)edge * C + c] * (L.scaler ? L.scaler[edge] : 1.0f) : L.base_w[edge];
    }
    for (int k = tid; k < L.I * (nk + 1); k += kSmallThreads) {
      const int i = k / (nk + 1), kk = k - i * (nk + 1);
      if (kk < nk) sK[L.kofs + k] = L.grid[(size_t)i * nk + kk];
    }
  }
  if (want_grads)
    for (int k = tid; k < M.trows * kSmallThreads; k += kSmallThreads) sT[k] = 0.0f;
  __syncthreads();
  for (int l = 0; l < M.nl; ++l) {             // 1/h of the equally spaced rows, from the staged knots (no global reads)
    const SmallLayer& L = M.L[l];
    const int nk = L.G + 7;
    if (tid < L.I) sK[L.kofs + tid * (nk + 1) + nk] = equal_spacing_inv_h(sK + L.kofs + tid * (nk + 1), nk);
  }
  __syncthreads();

  float* myA = sA + ps * M.slots;
  float* myG = sG + ps * kSmallWidth * 4;
  float* myC = sC + (tid - e) * 4;           // the contributions of this point's lanes
  float* myT = sT + tid;
  float loss_local = 0.0f;
  const float half_s2 = 0.5f * sigma * sigma;
  // (the lanes of a point sit in one wave: LDS accesses of a wave complete in program order, so the hand-offs through
  // sA / sG / sC below need no block barrier, only the compiler kept from reordering them)
#define PDEKAN_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
  for (int64_t p0 = (int64_t)blockIdx.x * PB; p0 < N; p0 += (int64_t)gridDim.x * PB) {
    const int64_t p = p0 + ps;
    const bool live = p < N;
    const int64_t pr = live ? p : N - 1;
    const float S = x[pr * sn], tt = x[pr * sn + sd];
    if (e < 2) *reinterpret_cast<float4*>(myA + e * 4) = (e == 0) ? make_float4(S, 1.0f, 0.0f, 0.0f) : make_float4(tt, 0.0f, 1.0f, 0.0f);
    PDEKAN_WAVE_SYNC();
    // ---- forward (the layer loops are unrolled over the most layers a model may have: the descriptors then sit in
    // scalar registers instead of being fetched from the kernel arguments point by point)
#pragma unroll
    for (int l = 0; l < kSmallLayersMax; ++l) {
      if (l >= M.nl) break;
      const SmallLayer& L = M.L[l];
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
      PDEKAN_WAVE_SYNC();
      if (e < L.O) {                          // lane e adds the I edges of output e, in order
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < L.I; ++k) {
          const float4 c = *reinterpret_cast<const float4*>(myC + (e * L.I + k) * 4);
          y.x += c.x; y.y += c.y; y.z += c.z; y.w += c.w;
        }
        *reinterpret_cast<float4*>(myA + L.aofs + L.I * 4 + e * 4) = y;
      }
      PDEKAN_WAVE_SYNC();
    }
    // ---- residual and loss (run1.py:128, :148): V = the one output of the last layer
    if (e == 0) {
      const float4 v = *reinterpret_cast<const float4*>(myA + M.slots - 4);
      const float cS = -rate * S, cSS = -half_s2 * S * S;
      const float res = v.z + cSS * v.w + cS * v.y + rate * v.x;
      const float diff = res - (target ? target[pr] : 0.0f);
      if (live) {
        if (res_out) res_out[p] = res;
        loss_local += diff * diff;
      }
      const float g = live ? 2.0f * diff * inv_n : 0.0f;
      *reinterpret_cast<float4*>(myG) = make_float4(g * rate, g * cS, g, g * cSS);
    }
    if (!want_grads) continue;
    PDEKAN_WAVE_SYNC();
    // ---- backward
#pragma unroll
    for (int lr = 0; lr < kSmallLayersMax; ++lr) {
      const int l = kSmallLayersMax - 1 - lr;
      if (l >= M.nl) continue;
      const SmallLayer& L = M.L[l];
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
        float* t = myT + (size_t)L.tofs * kSmallThreads;
        const float wb = w[C];
        float e1 = av.f1 * wb, e2 = av.f2 * wb, e3 = av.f3 * wb;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int c = bs.c0 + m;
          const bool in = edge_live && c >= 0 && c < C;
          const int cc = in ? c : 0;
          const float wv = in ? w[cc] : 0.0f;
          e1 += wv * bs.d1[m]; e2 += wv * bs.d2[m]; e3 += wv * bs.d3[m];
          if (in) t[cc * kSmallThreads] += A0 * bs.b[m] + A1 * bs.d1[m] + A2 * bs.d2[m];
        }
        if (edge_live) t[C * kSmallThreads] += A0 * av.f0 + A1 * av.f1 + A2 * av.f2;
        // y0 = phi, y1 = phi' a1, y2 = phi' a2, y3 = phi'' a1^2 + phi' a3
        *reinterpret_cast<float4*>(myC + e * 4) =
            make_float4(g.x * e1 + A1 * e2 + A2 * e3, g.y * e1 + 2.0f * g.w * a.y * e2, g.z * e1, g.w * e1);
      }
      if (l == 0) continue;                   // nobody asks for the gradient with respect to (S, t)
      PDEKAN_WAVE_SYNC();
      if (e < L.I) {                          // lane e adds the O edges that start at input e, in order
        float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < L.O; ++k) {
          const float4 c = *reinterpret_cast<const float4*>(myC + (k * L.I + e) * 4);
          ga.x += c.x; ga.y += c.y; ga.z += c.z; ga.w += c.w;
        }
        *reinterpret_cast<float4*>(myG + e * 4) = ga;
      }
      PDEKAN_WAVE_SYNC();
    }
  }
#undef PDEKAN_WAVE_SYNC
  // ---- the block's partial: the columns of the lanes that hold the same edge, added in a fixed order; the loss by a
  // fixed shuffle tree and the waves' sums in order
  float* dst = part + (size_t)blockIdx.x * (M.P + 1);
  __shared__ float s_loss[TB / 64];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) loss_local += __shfl_xor(loss_local, d);
  if (lane == 0) s_loss[tid >> 6] = loss_local;
  __syncthreads();
  if (tid == 0) {
    float t = 0.0f;
    for (int k = 0; k < TB / 64; ++k) t += s_loss[k];
    dst[M.P] = t;
  }
  if (want_grads) {
    for (int k = tid; k < M.P; k += kSmallThreads) {
      int l = 0;
      while (l + 1 < M.nl && k >= M.L[l + 1].wofs) ++l;
      const SmallLayer& L = M.L[l];
      const int C1 = L.G + 4;
      const int edge = (k - L.wofs) / C1, c = (k - L.wofs) - edge * C1;
      const float* row = sT + (size_t)(L.tofs + c) * kSmallThreads + edge;
      float sum = 0.0f;
#pragma unroll
      for (int q = 0; q < PB; ++q) sum += row[q * EP];
      dst[k] = sum;
    }
  }
}

// block = one edge (the last block: the loss), thread = (column c = tid & 31, slice s = tid >> 5): slice s adds the
// partials of blocks s, s + 32, ... (all loads of a thread independent: the partials were written by other CUs a moment
// ago and every load is a trip to memory), column c then adds its 32 slices in order, and the edge is unpacked
__global__ void __launch_bounds__(1024)
small_finish_kernel(SmallModel M, const float* __restrict__ part, int nblocks, float inv_n, float* __restrict__ loss_out,
                    int accumulate, int loss_accumulate, int want_grads) {
  __shared__ float s_sl[32][33];
  __shared__ float s_col[32];
  const int tid = threadIdx.x, c = tid & 31, sl = tid >> 5;
  const size_t stride = (size_t)M.P + 1;
  if (blockIdx.x == gridDim.x - 1) {            // the loss: thread t adds blocks t, t + 1024, ...
    float s = 0.0f;
    for (int b = tid; b < nblocks; b += 1024) s += part[(size_t)b * stride + M.P];
    s_sl[sl][c] = s;
    __syncthreads();
    if (tid < 32) {
      float t = 0.0f;
      for (int k = 0; k < 32; ++k) t += s_sl[k][tid];
      s_col[tid] = t;
    }
    __syncthreads();
    if (tid == 0 && loss_out) {
      float t = 0.0f;
      for (int k = 0; k < 32; ++k) t += s_col[k];
      t *= inv_n;
      *loss_out = loss_accumulate ? (*loss_out + t) : t;
    }
    return;
  }
  if (!want_grads) return;
  int l = 0, edge = blockIdx.x;
  while (l + 1 < M.nl && edge >= M.L[l].O * M.L[l].I) { edge -= M.L[l].O * M.L[l].I; ++l; }
  const SmallLayer& L = M.L[l];
  const int C = L.G + 3, C1 = C + 1;
  const bool col = c < C1;
  float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (col) {
    const float* src = part + L.wofs + edge * C1 + c;
    for (int b0 = sl; b0 < nblocks; b0 += 128) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + 32 * u;
        if (b < nblocks) v[u] += src[(size_t)b * stride];
      }
    }
  }
  s_sl[sl][c] = (v[0] + v[1]) + (v[2] + v[3]);
  __syncthreads();
  if (tid < 32) {
    float gw = 0.0f;
    for (int k = 0; k < 32; ++k) gw += s_sl[k][tid];
    s_col[tid] = (col && c < C) ? gw * L.spline_w[(size_t)edge * C + c] : 0.0f;
    if (col) {
      if (c < C) {
        float* d = L.g_spline + (size_t)edge * C + c;
        const float val = gw * (L.scaler ? L.scaler[edge] : 1.0f);
        *d = accumulate ? (*d + val) : val;
      } else {
        L.g_base[edge] = accumulate ? (L.g_base[edge] + gw) : gw;
      }
    }
  }
  __syncthreads();
  if (tid == 0 && L.scaler && L.g_scaler) {
    float gs = 0.0f;
    for (int k = 0; k < C; ++k) gs += s_col[k];
    L.g_scaler[edge] = accumulate ? (L.g_scaler[edge] + gs) : gs;
  }
}

static bool small_disabled() {               // A/B switch for tests and profiling
  const char* v = getenv("PDEKAN_DISABLE_SMALL_STEP");
  return v && v[0] == '1';
}

// lanes per point: the most edges a layer has, 16 or 64
static int small_lanes_per_point(const LayerView* layers, int nl) {
  int edges = 0;
  for (int l = 0; l < nl; ++l) edges = layers[l].O * layers[l].I > edges ? layers[l].O * layers[l].I : edges;
  return edges <= 16 ? 16 : 64;
}

struct SmallSizes { int P, kwords, slots, trows, ep, tb; size_t lds; };
static SmallSizes small_sizes(const LayerView* layers, int nl) {
  SmallSizes z = {0, 0, 0, 0, small_lanes_per_point(layers, nl), 0, 0};
  for (int l = 0; l < nl; ++l) {
    z.P += (int)layers[l].packed_floats();
    z.kwords += layers[l].I * (layers[l].nk() + 1);
    z.slots += layers[l].I * 4;
    z.trows += layers[l].C1();
  }
  z.slots += layers[nl - 1].O * 4;
  for (int tb : {kSmallThreadsMax, 256}) {
    const int pb = tb / z.ep;
    z.tb = tb;
    z.lds = ((size_t)((z.P + 3) & ~3) + ((z.kwords + 3) & ~3) + (size_t)pb * z.slots + (size_t)pb * kSmallWidth * 4 +
             (size_t)tb * 4 + (size_t)z.trows * tb) * sizeof(float);
    if (z.lds <= 150 * 1024) break;
  }
  return z;
}

// Models the two-launch step takes: (S, t) inputs, one output, at most kSmallLayersMax layers no wider than 8, grid
// size <= 20, and all tables within the LDS of one CU.  *lds (if given): dynamic LDS bytes of small_step_kernel.
bool small_step_eligible(const LayerView* layers, int nl, size_t* lds) {
  if (small_disabled() || nl < 1 || nl > kSmallLayersMax) return false;
  if (layers[0].I != 2 || layers[nl - 1].O != 1) return false;
  for (int l = 0; l < nl; ++l) {
    const LayerView& L = layers[l];
    if (L.I < 1 || L.I > kSmallWidth || L.O < 1 || L.O > kSmallWidth || L.G < 1 || L.G > 20) return false;
  }
  const SmallSizes z = small_sizes(layers, nl);
  if (lds) *lds = z.lds;
  return z.lds <= 150 * 1024;
}

static int small_points_per_block(const LayerView* layers, int nl) { const SmallSizes z = small_sizes(layers, nl); return z.tb / z.ep; }

static int small_step_blocks(const LayerView* layers, int nl, int64_t N) {
  const int pb = small_points_per_block(layers, nl);
  int64_t b = (N + pb - 1) / pb;
  if (b > kSmallBlocksMax) b = kSmallBlocksMax;
  return (int)(b < 1 ? 1 : b);
}

size_t small_step_part_floats(const LayerView* layers, int nl, int64_t N) {
  size_t P = 0;
  for (int l = 0; l < nl; ++l) P += layers[l].packed_floats();
  return (size_t)small_step_blocks(layers, nl, N) * (P + 1);
}

void launch_small_step(const LayerView* layers, const SmallGrads* grads, int nl, const float* x, int64_t sn, int64_t sd,
                       const float* target, int64_t N, float sigma, float rate, float inv_n, float* res_out,
                       float* loss_out, float* part, int accumulate, hipStream_t st) {
  SmallModel M;
  M.nl = nl;
  const SmallSizes z = small_sizes(layers, nl);
  int P = 0, kw = 0, slots = 0, trows = 0, edges = 0;
  for (int l = 0; l < nl; ++l) {
    SmallLayer& S = M.L[l];
    const LayerView& L = layers[l];
    S.I = L.I; S.O = L.O; S.G = L.G; S.act = L.act;
    S.grid = L.grid; S.base_w = L.base_w; S.spline_w = L.spline_w; S.scaler = L.scaler;
    S.g_base = grads ? grads[l].base_w : nullptr;
    S.g_spline = grads ? grads[l].spline_w : nullptr;
    S.g_scaler = grads ? grads[l].scaler : nullptr;
    S.wofs = P; S.kofs = kw; S.aofs = slots; S.tofs = trows;
    P += (int)L.packed_floats();
    kw += L.I * (L.nk() + 1);
    slots += L.I * 4;
    trows += L.C1();
    edges += L.O * L.I;
  }
  slots += layers[nl - 1].O * 4;
  M.P = P; M.kwords = kw; M.slots = slots; M.trows = trows;
  const int nblocks = small_step_blocks(layers, nl, N);
  const int want = grads ? 1 : 0;
  auto go = [&](auto kern) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), z.lds);
    kern<<<nblocks, z.tb, z.lds, st>>>(M, x, sn, sd, target, N, sigma, rate, inv_n, res_out, part, want);
  };
  if (z.ep == 16 && z.tb == 1024) go(small_step_kernel<16, 1024>);
  else if (z.ep == 16) go(small_step_kernel<16, 256>);
  else if (z.tb == 1024) go(small_step_kernel<64, 1024>);
  else go(small_step_kernel<64, 256>);
  small_finish_kernel<<<edges + 1, 1024, 0, st>>>(M, part, nblocks, inv_n, loss_out, accumulate, 0, want);
}

}  // namespace pdekan
