// One Black-Scholes training step of a SMALL KAN in two launches: models like the reference scripts' own
// (KAN([2, 5, 1]), the trunk KAN([2, 6, 1]) of run1.py:478-482; batch 100, run1.py:470-473), where a step of the
// layer-by-layer path is the sum of its launch latencies.
//
// Reference computation: KAN.forward (efficient_kan/kan.py:274-279, layers kan.py:153-166), its derivatives with
// respect to (S, t) (run1.py:121-124), the residual (run1.py:128), the squared-error loss (run1.py:148) and
// loss.backward() (run1.py:225).
//
//   small_step_kernel    thread = point.  The packed weights and knot rows of every layer sit in LDS; a thread carries
//                        the 4 jets of its point through the layers (activations of every layer boundary in its own
//                        LDS column: widths are run-time values), forms the residual, and walks back: input-jet
//                        gradients in registers, parameter gradients added into its OWN column of a gradient table in
//                        LDS (no two threads touch a word, so no atomics).  The block then adds the 64 columns in a
//                        fixed order and writes one partial gradient vector (+ its share of the loss) per block.
//   small_finish_kernel  adds the blocks' partials in a fixed order, unpacks them to the parameters' gradients
//                        (spline_weight: times the scaler; spline_scaler: sum over the columns of gW * spline_weight;
//                        kan.py:145-151 differentiated) and writes the loss.
// Every sum has a fixed order: results are bitwise reproducible.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kan_math.h"
#include "kan_launch.h"

namespace pdekan {

constexpr int kSmallThreads = 64;
constexpr int kSmallWidth = 8;               // widest layer boundary
constexpr int kSmallBlocksMax = 1024;

struct SmallLayer {
  int I, O, G, act;
  const float *grid, *base_w, *spline_w, *scaler;
  float *g_base, *g_spline, *g_scaler;
  int wofs;                                  // first packed weight of the layer: wp[o][i][C1], C1 = G + 4
  int kofs;                                  // first word of its knot rows in LDS: [i][nk + 1] (the last word: 1/h or 0)
  int aofs;                                  // first activation slot of its INPUT boundary: [feature][jet]
};
struct SmallModel {
  int nl, P, kwords, slots;                  // packed weights, knot words, activation slots (all boundaries)
  SmallLayer L[kSmallLayersMax];
};

// forward contribution of one edge: (phi, phi', phi'') from the 4 live bases and the base activation
__device__ __forceinline__ void edge_values(const float* w, int C, const Basis& bs, const Act& av, float& p0, float& p1,
                                            float& p2) {
  const float wb = w[C];
  p0 = av.f0 * wb; p1 = av.f1 * wb; p2 = av.f2 * wb;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = bs.c0 + m;
    const float wv = (c >= 0 && c < C) ? w[c] : 0.0f;
    p0 += wv * bs.b[m]; p1 += wv * bs.d1[m]; p2 += wv * bs.d2[m];
  }
}

__global__ void __launch_bounds__(kSmallThreads)
small_step_kernel(SmallModel M, const float* __restrict__ x, int64_t sn, int64_t sd, const float* __restrict__ target,
                  int64_t N, float sigma, float rate, float inv_n, float* __restrict__ res_out,
                  float* __restrict__ part, int want_grads) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;                          // [P]
  float* sK = sW + M.P;                      // knot rows
  float* sA = sK + M.kwords;                 // [slots][64]: this thread's activations / cotangents
  float* sT = sA + M.slots * kSmallThreads;  // [P][65]: this thread's parameter gradients
  const int tid = threadIdx.x;
  for (int l = 0; l < M.nl; ++l) {
    const SmallLayer& L = M.L[l];
    const int C = L.G + 3, C1 = C + 1, nk = L.G + 7;
    for (int e = tid; e < L.O * L.I * C1; e += kSmallThreads) {
      const int edge = e / C1, c = e - edge * C1;
      sW[L.wofs + e] = (c < C) ? L.spline_w[(size_t)edge * C + c] * (L.scaler ? L.scaler[edge] : 1.0f) : L.base_w[edge];
    }
    for (int e = tid; e < L.I * (nk + 1); e += kSmallThreads) {
      const int i = e / (nk + 1), k = e - i * (nk + 1);
      if (k < nk) sK[L.kofs + e] = L.grid[(size_t)i * nk + k];
    }
  }
  if (want_grads)
    for (int e = tid; e < M.P * 65; e += kSmallThreads) sT[e] = 0.0f;
  __syncthreads();
  for (int l = 0; l < M.nl; ++l) {             // 1/h of the equally spaced rows, from the staged knots (no global reads)
    const SmallLayer& L = M.L[l];
    const int nk = L.G + 7;
    if (tid < L.I) sK[L.kofs + tid * (nk + 1) + nk] = equal_spacing_inv_h(sK + L.kofs + tid * (nk + 1), nk);
  }
  __syncthreads();

  float* myA = sA + tid;
  float* myT = sT + tid;
  float loss_local = 0.0f;
  const float half_s2 = 0.5f * sigma * sigma;
  for (int64_t p0 = (int64_t)blockIdx.x * kSmallThreads; p0 < N; p0 += (int64_t)gridDim.x * kSmallThreads) {
    const int64_t p = p0 + tid;
    const bool live = p < N;
    const int64_t pr = live ? p : N - 1;
    const float S = x[pr * sn], tt = x[pr * sn + sd];
    // input jets of layer 0: (S, 1, 0, 0), (t, 0, 1, 0)
    myA[(0 * 4 + 0) * kSmallThreads] = S; myA[(0 * 4 + 1) * kSmallThreads] = 1.0f;
    myA[(0 * 4 + 2) * kSmallThreads] = 0.0f; myA[(0 * 4 + 3) * kSmallThreads] = 0.0f;
    myA[(1 * 4 + 0) * kSmallThreads] = tt; myA[(1 * 4 + 1) * kSmallThreads] = 0.0f;
    myA[(1 * 4 + 2) * kSmallThreads] = 1.0f; myA[(1 * 4 + 3) * kSmallThreads] = 0.0f;
    // ---- forward (the layer loops are unrolled over the most layers a model may have: the descriptors then sit in
    // scalar registers instead of being fetched from the kernel arguments point by point)
#pragma unroll
    for (int l = 0; l < kSmallLayersMax; ++l) {
      if (l >= M.nl) break;
      const SmallLayer& L = M.L[l];
      const int C = L.G + 3, C1 = C + 1, nk = L.G + 7;
      float y[kSmallWidth][4];
#pragma unroll
      for (int o = 0; o < kSmallWidth; ++o) { y[o][0] = y[o][1] = y[o][2] = y[o][3] = 0.0f; }
      for (int i = 0; i < L.I; ++i) {
        const float* a = myA + (L.aofs + i * 4) * kSmallThreads;
        const float a0 = a[0], a1 = a[kSmallThreads], a2 = a[2 * kSmallThreads], a3 = a[3 * kSmallThreads];
        Basis bs;
        eval_basis_row<false>(sK + L.kofs + i * (nk + 1), nk, a0, bs);
        const Act av = eval_act(L.act, a0);
        const float q = a1 * a1;
#pragma unroll
        for (int o = 0; o < kSmallWidth; ++o) {
          if (o < L.O) {
            float e0, e1, e2;
            edge_values(sW + L.wofs + (o * L.I + i) * C1, C, bs, av, e0, e1, e2);
            y[o][0] += e0; y[o][1] += e1 * a1; y[o][2] += e1 * a2; y[o][3] += e2 * q + e1 * a3;
          }
        }
      }
      float* dst = myA + (L.aofs + L.I * 4) * kSmallThreads;         // the next boundary
#pragma unroll
      for (int o = 0; o < kSmallWidth; ++o) {
        if (o < L.O) {
#pragma unroll
          for (int j = 0; j < 4; ++j) dst[(o * 4 + j) * kSmallThreads] = y[o][j];
        }
      }
    }
    // ---- residual and loss (run1.py:128, :148): V = the one output of the last layer
    float gy[kSmallWidth][4];
#pragma unroll
    for (int o = 0; o < kSmallWidth; ++o) { gy[o][0] = gy[o][1] = gy[o][2] = gy[o][3] = 0.0f; }
    {
      const float* v = myA + (M.slots - 4) * kSmallThreads;               // the last boundary: one output
      const float V = v[0], VS = v[kSmallThreads], Vt = v[2 * kSmallThreads], VSS = v[3 * kSmallThreads];
      const float cS = -rate * S, cSS = -half_s2 * S * S;
      const float res = Vt + cSS * VSS + cS * VS + rate * V;
      const float diff = res - (target ? target[pr] : 0.0f);
      if (live) {
        if (res_out) res_out[p] = res;
        loss_local += diff * diff;
      }
      const float g = live ? 2.0f * diff * inv_n : 0.0f;
      gy[0][0] = g * rate; gy[0][1] = g * cS; gy[0][2] = g; gy[0][3] = g * cSS;
    }
    if (!want_grads) continue;
    // ---- backward
#pragma unroll
    for (int lr = 0; lr < kSmallLayersMax; ++lr) {
      const int l = kSmallLayersMax - 1 - lr;
      if (l >= M.nl) continue;
      const SmallLayer& L = M.L[l];
      const int C = L.G + 3, C1 = C + 1, nk = L.G + 7;
      for (int i = 0; i < L.I; ++i) {
        float* a = myA + (L.aofs + i * 4) * kSmallThreads;
        const float a0 = a[0], a1 = a[kSmallThreads], a2 = a[2 * kSmallThreads], a3 = a[3 * kSmallThreads];
        Basis bs;
        eval_basis_row<true>(sK + L.kofs + i * (nk + 1), nk, a0, bs);
        const Act av = eval_act(L.act, a0);
        const float q = a1 * a1;
        float ga0 = 0.0f, ga1 = 0.0f, ga2 = 0.0f, ga3 = 0.0f;
#pragma unroll
        for (int o = 0; o < kSmallWidth; ++o) {
          if (o < L.O) {
            const int wi = L.wofs + (o * L.I + i) * C1;
            const float* w = sW + wi;
            const float g0 = gy[o][0], g1 = gy[o][1], g2 = gy[o][2], g3 = gy[o][3];
            // <gy, dy> = A0 phi + A1 phi' + A2 phi''
            const float A0 = g0, A1 = g1 * a1 + g2 * a2 + g3 * a3, A2 = g3 * q;
            float* t = myT + (size_t)wi * 65;
            const float wb = w[C];
            float e1 = av.f1 * wb, e2 = av.f2 * wb, e3 = av.f3 * wb;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int c = bs.c0 + m;
              const bool in = c >= 0 && c < C;
              const int cc = in ? c : 0;
              const float wv = in ? w[cc] : 0.0f;
              e1 += wv * bs.d1[m]; e2 += wv * bs.d2[m]; e3 += wv * bs.d3[m];
              if (in) t[cc * 65] += A0 * bs.b[m] + A1 * bs.d1[m] + A2 * bs.d2[m];
            }
            t[C * 65] += A0 * av.f0 + A1 * av.f1 + A2 * av.f2;
            // y0 = phi, y1 = phi' a1, y2 = phi' a2, y3 = phi'' a1^2 + phi' a3
            ga0 += g0 * e1 + (g1 * a1 + g2 * a2 + g3 * a3) * e2 + g3 * q * e3;
            ga1 += g1 * e1 + 2.0f * g3 * a1 * e2;
            ga2 += g2 * e1;
            ga3 += g3 * e1;
          }
        }
        a[0] = ga0; a[kSmallThreads] = ga1; a[2 * kSmallThreads] = ga2; a[3 * kSmallThreads] = ga3;
      }
      if (l > 0) {
#pragma unroll
        for (int o = 0; o < kSmallWidth; ++o) {
          if (o < L.I) {
            const float* a = myA + (L.aofs + o * 4) * kSmallThreads;
            gy[o][0] = a[0]; gy[o][1] = a[kSmallThreads]; gy[o][2] = a[2 * kSmallThreads]; gy[o][3] = a[3 * kSmallThreads];
          }
        }
      }
    }
  }
  // ---- the block's partial: columns added in a fixed order; the loss by a fixed shuffle tree
  float* dst = part + (size_t)blockIdx.x * (M.P + 1);
  if (want_grads) {
    __syncthreads();
    for (int e = tid; e < M.P; e += kSmallThreads) {
      const float* row = sT + (size_t)e * 65;
      float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll 4
      for (int k = 0; k < kSmallThreads; k += 4) { s0 += row[k]; s1 += row[k + 1]; s2 += row[k + 2]; s3 += row[k + 3]; }
      dst[e] = (s0 + s1) + (s2 + s3);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) loss_local += __shfl_xor(loss_local, d);
  if (tid == 0) dst[M.P] = loss_local;
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

// Models the two-launch step takes: (S, t) inputs, one output, at most kSmallLayersMax layers no wider than 8, grid
// size <= 20, and all tables within the LDS of one CU.  *lds (if given): dynamic LDS bytes of small_step_kernel.
bool small_step_eligible(const LayerView* layers, int nl, size_t* lds) {
  if (small_disabled() || nl < 1 || nl > kSmallLayersMax) return false;
  if (layers[0].I != 2 || layers[nl - 1].O != 1) return false;
  size_t P = 0, kw = 0, slots = 0;
  for (int l = 0; l < nl; ++l) {
    const LayerView& L = layers[l];
    if (L.I < 1 || L.I > kSmallWidth || L.O < 1 || L.O > kSmallWidth || L.G < 1 || L.G > 20) return false;
    P += L.packed_floats();
    kw += (size_t)L.I * (L.nk() + 1);
    slots += (size_t)L.I * 4;
  }
  slots += (size_t)layers[nl - 1].O * 4;
  const size_t bytes = (P + kw + slots * kSmallThreads + P * 65) * sizeof(float);
  if (lds) *lds = bytes;
  return bytes <= 150 * 1024;
}

int small_step_blocks(int64_t N) {
  int64_t b = (N + kSmallThreads - 1) / kSmallThreads;
  if (b > kSmallBlocksMax) b = kSmallBlocksMax;
  return (int)(b < 1 ? 1 : b);
}

size_t small_step_part_floats(const LayerView* layers, int nl, int64_t N) {
  size_t P = 0;
  for (int l = 0; l < nl; ++l) P += layers[l].packed_floats();
  return (size_t)small_step_blocks(N) * (P + 1);
}

void launch_small_step(const LayerView* layers, const SmallGrads* grads, int nl, const float* x, int64_t sn, int64_t sd,
                       const float* target, int64_t N, float sigma, float rate, float inv_n, float* res_out,
                       float* loss_out, float* part, int accumulate, hipStream_t st) {
  SmallModel M;
  M.nl = nl;
  int P = 0, kw = 0, slots = 0, edges = 0;
  for (int l = 0; l < nl; ++l) {
    SmallLayer& S = M.L[l];
    const LayerView& L = layers[l];
    S.I = L.I; S.O = L.O; S.G = L.G; S.act = L.act;
    S.grid = L.grid; S.base_w = L.base_w; S.spline_w = L.spline_w; S.scaler = L.scaler;
    S.g_base = grads ? grads[l].base_w : nullptr;
    S.g_spline = grads ? grads[l].spline_w : nullptr;
    S.g_scaler = grads ? grads[l].scaler : nullptr;
    S.wofs = P; S.kofs = kw; S.aofs = slots;
    P += (int)L.packed_floats();
    kw += L.I * (L.nk() + 1);
    slots += L.I * 4;
    edges += L.O * L.I;
  }
  slots += layers[nl - 1].O * 4;
  M.P = P; M.kwords = kw; M.slots = slots;
  size_t lds = 0;
  small_step_eligible(layers, nl, &lds);
  ensure_dynamic_lds(reinterpret_cast<const void*>(small_step_kernel), lds);
  const int nblocks = small_step_blocks(N);
  const int want = grads ? 1 : 0;
  small_step_kernel<<<nblocks, kSmallThreads, lds, st>>>(M, x, sn, sd, target, N, sigma, rate, inv_n, res_out, part, want);
  small_finish_kernel<<<edges + 1, 1024, 0, st>>>(M, part, nblocks, inv_n, loss_out, accumulate, 0, want);
}

}  // namespace pdekan
