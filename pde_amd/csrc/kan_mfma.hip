// Dense matrix-core path for wide hidden KAN layers on gfx950, J = 4 jets per
// scalar (value, d/dS, d/dt, d2/dS2): fp32 MFMA kernels for all three
// contractions, and a split-bf16 MFMA kernel (fp32 accuracy, see its own
// header below) for the weight gradients of layers with equally spaced knots.
//
// With the 4 jets of a point stacked as 4 matrix rows, one layer is three
// GEMM-shaped contractions over K = in_features * C1 expanded basis columns
// (C1 = G+3 spline bases + 1 base-activation column per input feature):
//   forward        Y [4N x O]  = E [4N x K] * W^T [K x O]
//   input grads    gE[4N x K]  = gY[4N x O] * W   [O x K]   (+ chain rule through the basis)
//   weight grads   gW[K x O]   = E^T[K x 4N] * gY [4N x O]
// E is never stored: every kernel rebuilds the slab of E it needs from the
// 16-byte input jet of a (point, feature) with a few dozen VALU ops, parks it
// in LDS in MFMA operand order, and the matrix cores do the rest in exact fp32
// (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: an fmaf chain, no reduced
// precision).  Only 4 of the G+3 spline columns of a feature are non-zero for
// a given point, so the E tile is zero-filled and 4+1 columns are overwritten.
//
// C/D register maps used below (dtype independent on gfx950):
//   32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
//   16x16: col = lane&15, row = 4*(lane>>4) + reg
// so a lane always owns 4 consecutive rows = the 4 jets of one point.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kan_math.h"
#include "kan_launch.h"

namespace pdekan {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kKnotPad = 28;   // >= G+7 knots + 1 slot for the feature's 1/h, G <= 20
constexpr int kKnotPadGen = 36;   // >= G+7 knots + 3 + 4 repeated end knots (bwd_weight_bf16x3_kernel, GEN)

// k-th float of the staged knot row of a feature: a knot, or (k == nk) its 1/h / 0
__device__ __forceinline__ float knot_row_value(const float* grid, const float* ginfo, int feature, int k, int nk) {
  return (k < nk) ? grid[(size_t)feature * nk + k] : (ginfo ? ginfo[feature] : 0.0f);
}

// ---- expanded basis jets of one (point, feature) -------------------------
// e[m] (m<4): jets of spline column c0+m ; e[4]: jets of the base-activation column
__device__ __forceinline__ void expand_jets(const float* knots, int nk, int act, float4 jet,
                                            float4 (&e)[5], int& c0) {
  Basis bs;
  eval_basis_row<false>(knots, nk, jet.x, bs);
  const Act av = eval_act(act, jet.x);
  const float s2 = jet.y * jet.y;
#pragma unroll
  for (int m = 0; m < 4; ++m)
    e[m] = make_float4(bs.b[m], bs.d1[m] * jet.y, bs.d1[m] * jet.z, bs.d2[m] * s2 + bs.d1[m] * jet.w);
  e[4] = make_float4(av.f0, av.f1 * jet.y, av.f1 * jet.z, av.f2 * s2 + av.f1 * jet.w);
  c0 = bs.c0;
}

// E tile layout: tile[col][RS] with the 4 jets of local point p at [col][4p .. 4p+3].
// A lane owns the same strip (feature slot f_local, point p_local) of the tile in
// every slab / pass, so after one full zero-fill it only has to clear the 4 columns
// it wrote last time and write the 4 live columns + the base column of this time:
// 9 unconditional 16-byte stores (a column outside [0, C) is redirected to the
// lane's private dump slot instead of being branched around).
template <int C1>
__device__ __forceinline__ void zero_strip(float* tile, int RS, int f_local, int p_local) {
  float* base = tile + (f_local * C1) * RS + p_local * 4;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < C1; ++c) *reinterpret_cast<float4*>(base + c * RS) = zero;
}

template <int C1>
__device__ __forceinline__ void write_expanded(float* tile, int RS, int f_local, int p_local,
                                               const float4 (&e)[5], int c0, int& prev_c0, float* dump) {
  constexpr int C = C1 - 1;
  float* base = tile + (f_local * C1) * RS + p_local * 4;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = prev_c0 + m;
    float* dst = (c >= 0 && c < C) ? base + c * RS : dump;
    *reinterpret_cast<float4*>(dst) = zero;
  }
  *reinterpret_cast<float4*>(base + C * RS) = e[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = c0 + m;
    float* dst = (c >= 0 && c < C) ? base + c * RS : dump;
    *reinterpret_cast<float4*>(dst) = e[m];
  }
  prev_c0 = c0;
}

// register <-> memory helpers (plain functions: arrays captured by a lambda end up in scratch)
template <int WV, int LIMIT>
__device__ __forceinline__ void fetch_rows16(f32x4 (&wreg)[WV], const float* src, int row_stride, int tid) {
  // LIMIT float4 in rows of 16 float4 (64 floats): idx -> row idx>>4, column 4*(idx&15)
#pragma unroll
  for (int v = 0; v < WV; ++v) {
    int idx = tid + v * 256;
    if (idx > LIMIT - 1) idx = LIMIT - 1;                // clamp: the load is unconditional, the store is not
    wreg[v] = *reinterpret_cast<const f32x4*>(src + (size_t)(idx >> 4) * row_stride + (idx & 15) * 4);
  }
}
template <int WV, int LIMIT>
__device__ __forceinline__ void store_rows16(const f32x4 (&wreg)[WV], float* dst, int tid) {
#pragma unroll
  for (int v = 0; v < WV; ++v) {
    const int idx = tid + v * 256;
    if (idx < LIMIT) *reinterpret_cast<f32x4*>(dst + (idx >> 4) * 64 + (idx & 15) * 4) = wreg[v];
  }
}

// ==========================================================================
// forward:  block = 4 waves, 64 points x 64 outputs; wave = 16 points (64 rows)
// K is walked in slabs of KI features (KK = KI*C1 columns, even).  The next
// slab's weights, knots and input jets are fetched into registers before the
// MFMA loop of the current slab, so their latency hides behind the matrix work;
// two barriers per slab.
// ==========================================================================
template <int C1, int KI>
__global__ void __launch_bounds__(256, 2)
fwd_mfma_kernel(const float4* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
                const float* __restrict__ wt,
                float4* __restrict__ y, int64_t N, int I, int O, int act) {
  constexpr int KK = KI * C1;
  static_assert(KK % 2 == 0, "k-step of 2");
  constexpr int nk = C1 + 3;               // G + 7
  constexpr int WV = (KK * 16 + 255) / 256;
  constexpr int kr = nk + 1;               // knots + 1/h
  static_assert(KI * kr <= 256, "one knot-row float per thread");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sW = reinterpret_cast<float*>(smem);            // [KK][64]
  float* sA = sW + KK * 64;                              // [4][KK][64]
  float* sK = sA + 4 * KK * 64;                          // [2][KI][kKnotPad]
  float* sD = sK + 2 * KI * kKnotPad;                    // [256][4] dump slots

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int p_local = lane & 15, f_local = lane >> 4;
  const int64_t P0 = (int64_t)blockIdx.x * 64 + wave * 16;
  const int o0 = blockIdx.y * 64;
  float* tileA = sA + wave * KK * 64;
  const int h = lane >> 5, r = lane & 31;
  int64_t pj = P0 + p_local;
  if (pj > N - 1) pj = N - 1;
  const int kf = tid / kr, kk_ = tid - kf * kr;          // knot-row slot of this thread

  f32x4 wreg[WV];                                        // native vectors: HIP float4 copies pin arrays in scratch
  float kreg = 0.f;
  float4 jet = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* wsrc = wt + o0;                           // slab i0 starts at row i0*C1
  const bool knot_thread = tid < KI * kr;
  const bool gen_lane = f_local < KI;

  if (knot_thread) sK[kf * kKnotPad + kk_] = knot_row_value(grid, ginfo, kf, kk_, nk);
  if (gen_lane) zero_strip<C1>(tileA, 64, f_local, p_local);
  int prev_c0 = -100;
  fetch_rows16<WV, KK * 16>(wreg, wsrc, O, tid);
  if (gen_lane) jet = a[pj * I + f_local];
  if (KI < I && knot_thread) kreg = knot_row_value(grid, ginfo, KI + kf, kk_, nk);

  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};

  int s = 0;
  for (int i0 = 0; i0 < I; i0 += KI, ++s) {
    __syncthreads();                                     // previous slab consumed; knots of this slab visible
    store_rows16<WV, KK * 16>(wreg, sW, tid);
    const bool more = i0 + KI < I;
    if (more && knot_thread) sK[((s + 1) & 1) * KI * kKnotPad + kf * kKnotPad + kk_] = kreg;
    if (gen_lane) {
      float4 e[5];
      int c0;
      expand_jets(sK + (s & 1) * KI * kKnotPad + f_local * kKnotPad, nk, act, jet, e, c0);
      write_expanded<C1>(tileA, 64, f_local, p_local, e, c0, prev_c0, sD + tid * 4);
    }
    if (more) {
      fetch_rows16<WV, KK * 16>(wreg, wsrc + (size_t)(i0 + KI) * C1 * O, O, tid);
      if (gen_lane) jet = a[pj * I + i0 + KI + f_local];
      if (i0 + 2 * KI < I && knot_thread) kreg = knot_row_value(grid, ginfo, i0 + 2 * KI + kf, kk_, nk);
    }
    __syncthreads();
    // operands are read from LDS one k-step ahead of the MFMAs that consume them
    float fa0[2], fa1[2], fb0[2], fb1[2];
    fa0[0] = tileA[h * 64 + r]; fa1[0] = tileA[h * 64 + 32 + r];
    fb0[0] = sW[h * 64 + r];    fb1[0] = sW[h * 64 + 32 + r];
#pragma unroll
    for (int t = 0; t < KK / 2; ++t) {
      const int cur = t & 1, nxt = cur ^ 1;
      if (t + 1 < KK / 2) {
        const int kk = 2 * (t + 1) + h;
        fa0[nxt] = tileA[kk * 64 + r]; fa1[nxt] = tileA[kk * 64 + 32 + r];
        fb0[nxt] = sW[kk * 64 + r];    fb1[nxt] = sW[kk * 64 + 32 + r];
      }
      __builtin_amdgcn_sched_barrier(0);                 // keep the reads ahead of the MFMAs
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cur], fb0[cur], acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cur], fb1[cur], acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cur], fb0[cur], acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cur], fb1[cur], acc11, 0, 0, 0);
    }
  }
  // rows 8g+4h .. +3 of an m-tile are the 4 jets of local point 2g+h (+8 per m-tile)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int64_t pa = P0 + 2 * g + h, pb = pa + 8;
    if (pa < N) {
      y[(pa * O + o0 + r)] = make_float4(acc00[4 * g], acc00[4 * g + 1], acc00[4 * g + 2], acc00[4 * g + 3]);
      y[(pa * O + o0 + 32 + r)] = make_float4(acc01[4 * g], acc01[4 * g + 1], acc01[4 * g + 2], acc01[4 * g + 3]);
    }
    if (pb < N) {
      y[(pb * O + o0 + r)] = make_float4(acc10[4 * g], acc10[4 * g + 1], acc10[4 * g + 2], acc10[4 * g + 3]);
      y[(pb * O + o0 + 32 + r)] = make_float4(acc11[4 * g], acc11[4 * g + 1], acc11[4 * g + 2], acc11[4 * g + 3]);
    }
  }
}

// ==========================================================================
// weight gradients: block = 4 waves; wave w owns KIW features (cols = KIW*C1,
// a multiple of 16) x 64 outputs of accumulators and the block walks its chunk
// of points PB at a time.  D[i = column][j = output] += E[row][column] * gY[row][output],
// one 16x16x4 MFMA reduces over the 4 jets of one point.  The gY tile and input
// jets of the next pass are fetched into registers before the MFMA loop.
// Output: part[chunk][O][I][C1] (summed over chunks by reduce_unpack).
// ==========================================================================
template <int C1, int KIW, int PB>
__global__ void __launch_bounds__(256, 2)
bwd_weight_mfma_kernel(const float4* __restrict__ a, const float4* __restrict__ gy,
                       const float* __restrict__ grid, const float* __restrict__ ginfo, float* __restrict__ part,
                       int64_t N, int I, int O, int act, int64_t chunk_pts, int nx, int ny, int nz, int skip_uniform) {
  constexpr int COLS = KIW * C1;
  static_assert(COLS % 16 == 0, "16-wide column tiles");
  static_assert(PB * KIW <= 64, "one (point, feature) per lane");
  constexpr int NT = COLS / 16;
  constexpr int ROWS = PB * 4;
  constexpr int RS = ROWS + 4;             // E tile row stride (16-byte aligned, not a multiple of 32)
  constexpr int RG = 80;                   // gY tile row stride: quarter-waves land on disjoint banks
  constexpr int nk = C1 + 3;
  constexpr int GV = (PB * 64 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sE = reinterpret_cast<float*>(smem);            // [4][COLS][RS]
  float* sG = sE + 4 * COLS * RS;                        // [ROWS][RG]
  float* sK = sG + ROWS * RG;                            // [4*KIW][kKnotPad]

  // 1-D launch, decoded so that the blocks which read the same gY rows (same point chunk and output
  // tile, different feature groups) are 8 block ids apart: consecutive ids go round-robin over the 8 XCDs,
  // so those blocks share one XCD's L2 and gY comes from HBM once.
  int bx, by, bz;
  {
    const int b = blockIdx.x;
    if (nz % 8 == 0) {
      const int r = b >> 3;
      bx = r % nx;
      by = (r / nx) % ny;
      bz = (r / (nx * ny)) * 8 + (b & 7);
    } else {
      bx = b % nx;
      by = (b / nx) % ny;
      bz = b / (nx * ny);
    }
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // blocks whose features all have (nearly) equally spaced knots belong to bwd_weight_bf16x3_kernel when it runs too
  if (skip_uniform == 1 && __builtin_amdgcn_ballot_w64(ginfo[bx * 4 * KIW + lane % (4 * KIW)] != 0.0f) == ~0ull) return;
  if (skip_uniform == 2 && all_rows_near_uniform(ginfo, I)) return;      // a 12-jet matrix kernel does the whole layer
  const int p_local = lane % PB, f_local = lane / PB;
  const bool gen_lane = f_local < KIW;
  const int i0 = (bx * 4 + wave) * KIW;
  const int o0 = by * 64;
  float* tileE = sE + wave * COLS * RS;
  const int q = lane >> 4, l15 = lane & 15;

  for (int idx = tid; idx < 4 * KIW * (nk + 1); idx += 256) {
    const int f = idx / (nk + 1), k = idx - f * (nk + 1);
    sK[f * kKnotPad + k] = knot_row_value(grid, ginfo, bx * 4 * KIW + f, k, nk);
  }
  if (gen_lane) zero_strip<C1>(tileE, RS, f_local, p_local);
  int prev_c0 = -100;
  f32x4 acc[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t pbeg = (int64_t)bz * chunk_pts;
  const int64_t pend = (pbeg + chunk_pts < N) ? pbeg + chunk_pts : N;

  float4 greg[GV];
  float4 jet = make_float4(0.f, 0.f, 0.f, 0.f);
#define PDEKAN_BWW_FETCH(PBASE)                                                          \
  {                                                                                      \
    _Pragma("unroll") for (int v = 0; v < GV; ++v) {                                     \
      const int idx = tid + v * 256;                                                     \
      const int p = idx >> 6, o = idx & 63;                                              \
      greg[v] = make_float4(0.f, 0.f, 0.f, 0.f);                                         \
      if (idx < PB * 64 && (PBASE) + p < pend) greg[v] = gy[((PBASE) + p) * O + o0 + o]; \
    }                                                                                    \
    if (gen_lane) {                                                                      \
      int64_t pc = (PBASE) + p_local;                                                    \
      if (pc > N - 1) pc = N - 1;                                                        \
      jet = a[pc * I + i0 + f_local];                                                    \
    }                                                                                    \
  }
  if (pbeg < pend) PDEKAN_BWW_FETCH(pbeg)

  for (int64_t pb = pbeg; pb < pend; pb += PB) {
    __syncthreads();                                     // previous pass consumed (and knots visible)
#pragma unroll
    for (int v = 0; v < GV; ++v) {
      const int idx = tid + v * 256;
      if (idx < PB * 64) {
        const int p = idx >> 6, o = idx & 63;
        float* dst = sG + (p * 4) * RG + o;
        dst[0] = greg[v].x; dst[RG] = greg[v].y; dst[2 * RG] = greg[v].z; dst[3 * RG] = greg[v].w;
      }
    }
    if (gen_lane) {
      float4 e[5];
      int c0;
      expand_jets(sK + (wave * KIW + f_local) * kKnotPad, nk, act, jet, e, c0);
      // dump slot: the 4 pad floats that end the first row of this lane's strip
      write_expanded<C1>(tileE, RS, f_local, p_local, e, c0, prev_c0, tileE + (f_local * C1) * RS + ROWS);
    }
    if (pb + PB < pend) PDEKAN_BWW_FETCH(pb + PB)
    __syncthreads();
    // operands of point t+1 are read while the MFMAs of point t run
    float fb[2][4], fe[2][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m) fb[0][m] = sG[q * RG + m * 16 + l15];
#pragma unroll
    for (int n = 0; n < NT; ++n) fe[0][n] = tileE[(n * 16 + l15) * RS + q];
#pragma unroll
    for (int t = 0; t < PB; ++t) {
      const int cur = t & 1, nxt = cur ^ 1;
      if (t + 1 < PB) {
#pragma unroll
        for (int m = 0; m < 4; ++m) fb[nxt][m] = sG[(4 * (t + 1) + q) * RG + m * 16 + l15];
#pragma unroll
        for (int n = 0; n < NT; ++n) fe[nxt][n] = tileE[(n * 16 + l15) * RS + 4 * (t + 1) + q];
      }
      __builtin_amdgcn_sched_barrier(0);                 // keep the reads ahead of the MFMAs
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fe[cur][n], fb[cur][m], acc[n][m], 0, 0, 0);
    }
  }
#undef PDEKAN_BWW_FETCH
  // D[i = column][j = output]: this lane holds columns n*16 + 4q + reg of output m*16 + l15
  float* dst0 = part + ((size_t)bz * O + o0) * I * C1 + (size_t)i0 * C1;
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float* dst = dst0 + (size_t)(m * 16 + l15) * I * C1 + n * 16 + 4 * q;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) dst[rg] = acc[n][m][rg];
    }
}

// ==========================================================================
// weight gradients on the bf16 matrix cores with fp32 accuracy, for feature
// blocks whose knots are equally spaced (1/h known per feature).
//
// Every fp32 operand x is cut into three bf16 pieces hi + mid + lo by
// truncation (hi = the top 8 significant bits, mid the next 8, lo the last 8:
// the remainders are exact fp32 subtractions, so hi + mid + lo == x exactly)
// and a product a*b is taken as
//     hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi,
// six v_mfma_f32_16x16x32_bf16 whose bf16 x bf16 products are exact in the fp32
// accumulator.  The terms left out are below 2^-21 |a b| (2^-23 on average), the
// order of an fp32 product's own rounding, and the matrix work costs 6/16 of
// the fp32-MFMA time.
//
// Every wave is on its own (no block barrier): wave = FE features x C1 columns
// (NT = FE*C1/16 column tiles: 8 x 14 for grid 10, 4 x 24 for grid 20) x 64
// outputs of accumulators over a chunk of points, walked 8 points (K = 8 x 4
// jets = 32 = one MFMA) at a time.  One wave
// per SIMD, so the instruction stream itself is software-pipelined: the vector
// work that prepares step s+1 sits between the matrix instructions of step s.
//   B operand (gY): lane (r = lane&15, q = lane>>4) holds output 16m + r, jets of
//     points 2q and 2q+1: two 16-byte loads per output tile, fetched two steps
//     ahead into registers, cut into planes at the start of their step.
//   A operand (E^T): lane (p = lane&7, f = lane>>3) expands its (point,
//     feature) pair (the 4 live spline columns + the base column), cuts the 20
//     jets into planes and parks them in the wave's own LDS tile
//     [3 planes][8*C1 columns][32 bf16 + pad]; the 4 columns it wrote one step
//     earlier are cleared, everything else of the tile stays zero.  The tile is
//     read back in MFMA operand order (ds_read_b128).  LDS operations of a
//     wave complete in order, so the tile of step s+1 is written right after
//     the last operand read of step s has been issued.
// Two variants share everything but the basis evaluation: GEN = false for knot
// rows that are equally spaced, GEN = true for rows that are only nearly so
// (knot_spacing_info: the default grid of grid_size 20).  Both are launched on
// the same grid; a block (4 waves = 4*FE features) returns at once unless its
// features are all of the variant's kind (one row that is only nearly equally
// spaced sends the block to GEN = true).  bwd_weight_mfma_kernel, launched with
// skip_uniform = 1 on the same grid, does exactly the blocks both leave out.
// Output: part[chunk][O][I][C1].
// ==========================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// (x0, x1) -> one dword of each of the three bf16 planes, x0 in the low half
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  constexpr unsigned kHiHalves = 0x07060302u;            // v_perm_b32: {S0.hi16, S1.hi16}
  const unsigned a0 = __float_as_uint(x0), a1 = __float_as_uint(x1);
  h = __builtin_amdgcn_perm(a1, a0, kHiHalves);
  const float r0 = x0 - __uint_as_float(a0 & 0xffff0000u), r1 = x1 - __uint_as_float(a1 & 0xffff0000u);
  const unsigned b0 = __float_as_uint(r0), b1 = __float_as_uint(r1);
  m = __builtin_amdgcn_perm(b1, b0, kHiHalves);
  const float s0 = r0 - __uint_as_float(b0 & 0xffff0000u), s1 = r1 - __uint_as_float(b1 & 0xffff0000u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), kHiHalves);
}

struct Planes2 { u32x2 h, m, l; };                         // the 4 jets of one scalar
struct GyPlanes { u32x4 h[4], m[4], l[4]; };               // 4 output tiles x (2 points x 4 jets)

__device__ __forceinline__ Planes2 split_jet(float4 v) {
  Planes2 r;
  unsigned h0, m0, l0, h1, m1, l1;
  split_pair(v.x, v.y, h0, m0, l0);
  split_pair(v.z, v.w, h1, m1, l1);
  r.h = (u32x2){h0, h1}; r.m = (u32x2){m0, m1}; r.l = (u32x2){l0, l1};
  return r;
}

__device__ __forceinline__ void split_gy_tile(const float4 (&g)[2], GyPlanes& G, int m) {
  const Planes2 p0 = split_jet(g[0]), p1 = split_jet(g[1]);
  G.h[m] = (u32x4){p0.h.x, p0.h.y, p1.h.x, p1.h.y};
  G.m[m] = (u32x4){p0.m.x, p0.m.y, p1.m.x, p1.m.y};
  G.l[m] = (u32x4){p0.l.x, p0.l.y, p1.l.x, p1.l.y};
}

// the expanded jets of one (point, feature) pair: 4 live spline columns c0..c0+3 and the base column
struct PairJets { float4 col[5]; int c0; };
struct PairPlanes { Planes2 col[5]; };

struct FeatureGrid { float g0, glast, inv_h; };            // of this lane's feature: constant over the kernel

// GEN = false: knots equally spaced (fixed polynomial pieces); GEN = true: nearly so (general triangle on the
// knots around the quotient guess; the row is stored with repeated end knots, see eval_basis_near_uniform)
template <int ACT, bool GEN>
__device__ __forceinline__ void expand_pair(const float* knots, int nk, const FeatureGrid& fg, float4 jet, PairJets& out) {
  Basis bs;
  if constexpr (GEN) eval_basis_near_uniform(knots, nk, fg.g0, fg.glast, fg.inv_h, jet.x, bs);
  else eval_basis_uniform_by_quotient(knots, nk, fg.g0, fg.glast, fg.inv_h, jet.x, bs);
  const Act av = eval_act(ACT, jet.x);
  const float s2 = jet.y * jet.y;
#pragma unroll
  for (int m = 0; m < 4; ++m)
    out.col[m] = make_float4(bs.b[m], bs.d1[m] * jet.y, bs.d1[m] * jet.z, bs.d2[m] * s2 + bs.d1[m] * jet.w);
  out.col[4] = make_float4(av.f0, av.f1 * jet.y, av.f1 * jet.z, av.f2 * s2 + av.f1 * jet.w);
  out.c0 = bs.c0;
}

// park a pair's planes in the wave's E tile: clear the columns written one step earlier, write the live ones
template <int C1, int COLS, int RW>
__device__ __forceinline__ void park_pair(unsigned* base, unsigned* dump, const PairPlanes& pp, int c0, int& prev_c0) {
  constexpr int C = C1 - 1, PL = COLS * RW;
  const u32x2 zero = {0u, 0u};
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = prev_c0 + m;
    unsigned* dst = (c >= 0 && c < C) ? base + c * RW : dump;
    *reinterpret_cast<u32x2*>(dst) = zero;
    *reinterpret_cast<u32x2*>(dst + PL) = zero;
    *reinterpret_cast<u32x2*>(dst + 2 * PL) = zero;
  }
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const int c = (m < 4) ? c0 + m : C;
    unsigned* dst = (m == 4 || (c >= 0 && c < C)) ? base + c * RW : dump;
    *reinterpret_cast<u32x2*>(dst) = pp.col[m].h;
    *reinterpret_cast<u32x2*>(dst + PL) = pp.col[m].m;
    *reinterpret_cast<u32x2*>(dst + 2 * PL) = pp.col[m].l;
  }
  prev_c0 = c0;
}

struct TileOps { u32x4 h, m, l; };

template <int COLS, int RW>
__device__ __forceinline__ TileOps read_tile(const unsigned* esrc, int t) {
  TileOps o;
  o.h = *reinterpret_cast<const u32x4*>(esrc + t * 16 * RW);
  o.m = *reinterpret_cast<const u32x4*>(esrc + t * 16 * RW + COLS * RW);
  o.l = *reinterpret_cast<const u32x4*>(esrc + t * 16 * RW + 2 * COLS * RW);
  return o;
}

__device__ __forceinline__ void mfma_tile(f32x4 (&acc)[4], const TileOps& e, const GyPlanes& G) {
  const bf16x8 eh = __builtin_bit_cast(bf16x8, e.h), em = __builtin_bit_cast(bf16x8, e.m),
               el = __builtin_bit_cast(bf16x8, e.l);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const bf16x8 bh = __builtin_bit_cast(bf16x8, G.h[m]), bm = __builtin_bit_cast(bf16x8, G.m[m]),
                 bl = __builtin_bit_cast(bf16x8, G.l[m]);
    f32x4 c = acc[m];                        // small terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(el, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eh, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(em, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(em, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eh, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eh, bh, c, 0, 0, 0);
    acc[m] = c;
  }
}

// 16 bytes at a wave-uniform row address plus a 32-bit per-lane byte offset (scalar base + vector offset load)
__device__ __forceinline__ float4 ld16(const char* __restrict__ row, unsigned off) {
  return *reinterpret_cast<const float4*>(row + off);
}

constexpr int kBf16FlushPoints = 512;        // points accumulated by the matrix instructions between two vector adds
constexpr int kBf16Rw = 20;                  // words per E tile row: 32 bf16 + 16 bytes of pad (bank spread; dump slot)

// One step = 8 points = NT regions of 24 matrix instructions (one column tile each; NT = 7 is described, NT = 6
// moves everything from region 3 on one region earlier), fenced so that each
// region keeps its own share of the vector work that prepares step s+1 (one wave per SIMD: nothing else would
// fill the matrix instructions' issue gaps).  graw / xraw hold the raw rows of step s+1 on entry; each piece is
// fetched again (rows of step s+2, at pnext2) right after the region that consumed it:
//   region 0        evaluate the basis of the pairs of step s+1 (xraw); fetch gY tile 3 of step s+1 (pnext1)
//   region 1        fetch xraw of step s+2
//   regions 1..3    cut the pairs' 5 columns into planes
//   regions 3..6    cut gY tile 0..3 of step s+1 into planes (Gn); regions 4..6 fetch tiles 0..2 of step s+2
//   region 4        read the operands of tiles 5 and 6, then park the pairs' planes of step s+1
//   region 6        read the first operands of step s+1
template <int C1, int FE, int ACT, bool GEN>
__device__ __forceinline__ void bf16x3_step(f32x4 (&acc)[FE * C1 / 16][4], const GyPlanes& G, GyPlanes& Gn, TileOps& cur,
                                            float4 (&graw)[4][2], float4& xraw, unsigned* sE, const float* knots,
                                            const FeatureGrid& fg, int& prev_c0, const char* __restrict__ xrow2, const char* __restrict__ grow1,
                                            const char* __restrict__ grow2, unsigned xoff, unsigned goff, int lane,
                                            int O) {
  constexpr int COLS = FE * C1, NT = COLS / 16, RW = kBf16Rw, nk = C1 + 3;
  static_assert(NT == 6 || NT == 7, "the region plan below is written for 6 or 7 column tiles");
  const int r = lane & 15, q = lane >> 4, p_local = lane & 7, f_local = (lane >> 3) % FE;
  const unsigned* esrc = sE + r * RW + q * 4;
  PairJets pj;
  PairPlanes pp;
  TileOps last;                              // operands of tile 6, read one region early so that the park can go early
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    TileOps nxt;
    if (t + 1 < NT - 1) nxt = read_tile<COLS, RW>(esrc, t + 1);
    if (t == NT - 3) last = read_tile<COLS, RW>(esrc, NT - 1);
    if (t == NT - 2) nxt = last;
    if (t == 0) {
      expand_pair<ACT, GEN>(knots, nk, fg, xraw, pj);
#pragma unroll
      for (int e = 0; e < 2; ++e) graw[3][e] = ld16(grow1, goff + (unsigned)(e * O + 3 * 16) * 16u);
    }
    if (t == 1) xraw = ld16(xrow2, xoff);
    if (t == 1) { pp.col[0] = split_jet(pj.col[0]); pp.col[1] = split_jet(pj.col[1]); }
    if (t == 2) { pp.col[2] = split_jet(pj.col[2]); pp.col[3] = split_jet(pj.col[3]); }
    if (t == NT - 4) pp.col[4] = split_jet(pj.col[4]);
    if (t >= NT - 4) split_gy_tile(graw[t - (NT - 4)], Gn, t - (NT - 4));
    if (t >= NT - 3) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
        graw[t - (NT - 3)][e] = ld16(grow2, goff + (unsigned)(e * O + (t - (NT - 3)) * 16) * 16u);
    }
    if (t == NT - 3) {                       // every operand read of this step's E tile has been issued
      unsigned* base = sE + (f_local * C1) * RW + p_local * 2;
      park_pair<C1, COLS, RW>(base, sE + (f_local * C1) * RW + 16, pp, pj.c0, prev_c0);
    }
    if (t == NT - 1) nxt = read_tile<COLS, RW>(esrc, 0);
    mfma_tile(acc[t], cur, G);
    cur = nxt;
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int C1, int FE, int ACT, bool GEN>
__global__ void __launch_bounds__(256, 1)
bwd_weight_bf16x3_kernel(const float4* __restrict__ a, const float4* __restrict__ gy,
                         const float* __restrict__ grid, const float* __restrict__ ginfo, float* __restrict__ part,
                         int64_t N, int I, int O, int64_t chunk_pts, int nx, int ny, int nz) {
  constexpr int COLS = FE * C1, NT = COLS / 16, RW = kBf16Rw, nk = C1 + 3;
  static_assert(COLS % 16 == 0 && (FE == 8 || FE == 4), "16-wide column tiles; 8 points x FE features per step");
  static_assert(nk + 1 <= kKnotPad, "knot row + 1/h fit the padded row");
  constexpr int KP = GEN ? kKnotPadGen : kKnotPad;        // floats per staged knot row
  static_assert(nk + 7 <= kKnotPadGen, "knot row with 3 + 4 repeated end knots fits");
  constexpr int kWaveWords = 3 * COLS * RW + FE * KP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bx, by, bz;                            // same decode as bwd_weight_mfma_kernel
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
  {                                          // the block's features: all equally spaced -> GEN = false takes it;
    const float gi = ginfo[bx * 4 * FE + lane % (4 * FE)];          // all at least nearly so -> GEN = true
    const bool all_near = __builtin_amdgcn_ballot_w64(gi != 0.0f) == ~0ull;
    const bool all_equal = __builtin_amdgcn_ballot_w64(gi > 0.0f) == ~0ull;
    if (!all_near || all_equal == GEN) return;
  }

  unsigned* sE = reinterpret_cast<unsigned*>(smem) + wave * kWaveWords;   // this wave's tile: [3][COLS][RW]
  float* sK = reinterpret_cast<float*>(sE + 3 * COLS * RW);               // and its FE knot rows
  // FE = 4: lanes 32..63 repeat the pairs of lanes 0..31 (same values to the same LDS words)
  const int r = lane & 15, q = lane >> 4, p_local = lane & 7, f_local = (lane >> 3) % FE;
  const int i0 = (bx * 4 + wave) * FE, o0 = by * 64;
  if constexpr (GEN) {                       // [g_0 x3][the nk knots][g_last x4]
    for (int idx = lane; idx < FE * KP; idx += 64) {
      const int f = idx / KP, k = idx - f * KP - 3;
      sK[idx] = grid[(size_t)(i0 + f) * nk + (k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k))];
    }
  } else {
    for (int idx = lane; idx < FE * (nk + 1); idx += 64) {
      const int f = idx / (nk + 1), k = idx - f * (nk + 1);
      sK[f * KP + k] = knot_row_value(grid, ginfo, i0 + f, k, nk);
    }
  }
  for (int idx = lane; idx < 3 * COLS * RW; idx += 64) sE[idx] = 0u;
  const float* knots = sK + f_local * KP + (GEN ? 3 : 0);
  const FeatureGrid fg = {grid[(size_t)(i0 + f_local) * nk], grid[(size_t)(i0 + f_local) * nk + nk - 1],
                          fabsf(ginfo[i0 + f_local])};

  // The matrix instruction's accumulate is not round-to-nearest: over a run of steps the running sums drift by an
  // amount that grows with the length of the run (measured against fp64: 1.6e-5 of the gradient after 4096 points,
  // 1.7e-6 after 512).  So the matrix instructions accumulate kBf16FlushPoints points at a time into acc, which
  // is then added (vector add, round-to-nearest) into tot and cleared; tot is touched nowhere else.
  f32x4 acc[NT][4], tot[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) { acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f}; tot[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  // N and chunk_pts are multiples of 16: whole pairs of steps
  const int64_t pbeg = (int64_t)bz * chunk_pts;
  const int64_t pend = (pbeg + chunk_pts < N) ? pbeg + chunk_pts : N;
  // rows are addressed as (wave-uniform row pointer) + (32-bit lane offset)
  const char* xb = reinterpret_cast<const char*>(a);
  const char* gb = reinterpret_cast<const char*>(gy);
  const unsigned xoff = (unsigned)(p_local * I + i0 + f_local) * 16u;
  const unsigned goff = (unsigned)(2 * q * O + o0 + r) * 16u;
  const size_t xstride = (size_t)I * 16, gstride = (size_t)O * 16;       // bytes per point

  if (pbeg < pend) {
    float4 graw[4][2], xraw;
    GyPlanes GA, GB;
    int prev_c0 = -100;
    {
      float4 g0[4][2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          g0[m][e] = ld16(gb + pbeg * gstride, goff + (unsigned)(e * O + m * 16) * 16u);
          graw[m][e] = ld16(gb + (pbeg + 8) * gstride, goff + (unsigned)(e * O + m * 16) * 16u);
        }
      const float4 x0 = ld16(xb + pbeg * xstride, xoff);
      xraw = ld16(xb + (pbeg + 8) * xstride, xoff);
      PairJets pj;
      PairPlanes pp;
      expand_pair<ACT, GEN>(knots, nk, fg, x0, pj);
#pragma unroll
      for (int m = 0; m < 5; ++m) pp.col[m] = split_jet(pj.col[m]);
      unsigned* base = sE + (f_local * C1) * RW + p_local * 2;
      park_pair<C1, COLS, RW>(base, sE + (f_local * C1) * RW + 16, pp, pj.c0, prev_c0);
#pragma unroll
      for (int m = 0; m < 4; ++m) split_gy_tile(g0[m], GA, m);
    }
    TileOps cur = read_tile<COLS, RW>(sE + r * RW + q * 4, 0);
    for (int64_t pb = pbeg; pb < pend;) {
      const int64_t pstop = (pb + kBf16FlushPoints < pend) ? pb + kBf16FlushPoints : pend;
      for (; pb < pstop; pb += 16) {
        // past the end of the chunk: any valid rows, fetched (and expanded) but never used
        const int64_t p1 = pb + 8, p2 = (pb + 16 < pend) ? pb + 16 : pbeg, p3 = (pb + 24 < pend) ? pb + 24 : pbeg;
        bf16x3_step<C1, FE, ACT, GEN>(acc, GA, GB, cur, graw, xraw, sE, knots, fg, prev_c0, xb + p2 * xstride,
                                      gb + p1 * gstride, gb + p2 * gstride, xoff, goff, lane, O);
        bf16x3_step<C1, FE, ACT, GEN>(acc, GB, GA, cur, graw, xraw, sE, knots, fg, prev_c0, xb + p3 * xstride,
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
      *reinterpret_cast<f32x4*>(dst) = tot[t][m];
    }
}

// ==========================================================================
// weight gradients of a layer with KI = 2 inputs (the first layer of a PINN on
// (S, t)): all KI*C1 expanded columns fit two or three 16-wide tiles, so every
// wave owns ALL columns x 64 outputs and the 4 waves of a block split the
// points instead (8 per pass and wave, no block barrier inside the loop); their
// accumulators are added through LDS at the end.  The E tile is built as in the
// kernel above; the gY operand needs no LDS: the 16x16x4 instruction contracts
// over 4 points of one jet slot, so the B operands of a lane (n = lane&15,
// k = lane>>4) for the four slots are the 16 bytes gY[point k][output n][0..3].
// Output: part[chunk][O][KI][C1].
// ==========================================================================
template <int C1, int KI>
__global__ void __launch_bounds__(256, 2)
bwd_weight_mfma_skinny_kernel(const float4* __restrict__ a, const float* __restrict__ gy,
                              const float* __restrict__ grid, const float* __restrict__ ginfo,
                              float* __restrict__ part, int64_t N, int O, int act, int64_t chunk_pts) {
  constexpr int COLS = KI * C1, NT = (COLS + 15) / 16, PB = 8, ROWS = PB * 4, RS = ROWS + 4;
  constexpr int nk = C1 + 3;
  static_assert(PB * KI <= 64, "one (point, feature) per lane");
  static_assert(RS <= 64, "the E tiles fit the area later used for the final reduction");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sE = reinterpret_cast<float*>(smem);            // [4][NT*16][RS]; afterwards [4][NT*16][64] partial sums
  float* sK = sE + 4 * NT * 16 * 64;                     // [KI][kKnotPad]   (64 >= RS: room for both uses)

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int p_local = lane % PB, f_local = lane / PB;
  const bool gen_lane = f_local < KI;
  const int o0 = blockIdx.y * 64;
  float* tileE = sE + wave * NT * 16 * RS;
  const int q = lane >> 4, l15 = lane & 15;

  for (int idx = tid; idx < KI * (nk + 1); idx += 256) {
    const int f = idx / (nk + 1), k = idx - f * (nk + 1);
    sK[f * kKnotPad + k] = knot_row_value(grid, ginfo, f, k, nk);
  }
  for (int idx = lane; idx < NT * 16 * RS; idx += 64) tileE[idx] = 0.0f;     // pad columns stay zero
  __syncthreads();
  int prev_c0 = -100;
  f32x4 acc[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t pbeg = (int64_t)blockIdx.x * chunk_pts;
  const int64_t pend = (pbeg + chunk_pts < N) ? pbeg + chunk_pts : N;
  // The matrix instruction contracts over 4 POINTS for one jet slot at a time: lane (n = lane&15, k = lane>>4)
  // supplies gY[point 4*t4 + k][output n][jet j] for B and E[column n][point 4*t4 + k][jet j] for A, j = 0..3 in
  // four instructions.  The four B operands of a lane are then one 16-byte load (all jets of one output of one
  // point) and the four A operands one 16-byte LDS read.
  const int loff = (q * O + o0 + l15) * 4;               // this lane's first float inside a group of 4 point rows of gy

  f32x4 bA[2][4], bB[2][4];                              // [group of 4 points][output tile]
  float4 jet = make_float4(0.f, 0.f, 0.f, 0.f);
  // (PBASE is wave-uniform: row addresses are scalar, only `loff` is per lane)
#define PDEKAN_SKINNY_FETCH(BUF, PBASE)                                                  \
  {                                                                                      \
    if ((PBASE) + PB <= pend) {                                                          \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                    \
        const float* row = gy + (size_t)((PBASE) + 4 * t) * O * 4;                       \
        _Pragma("unroll") for (int m = 0; m < 4; ++m)                                    \
          BUF[t][m] = *reinterpret_cast<const f32x4*>(row + loff + m * 64);              \
      }                                                                                  \
    } else {                                                                             \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                    \
        const bool in = (PBASE) + 4 * t + q < pend;                                      \
        const float* row = gy + ((size_t)(in ? (PBASE) + 4 * t + q : pbeg) * O + o0 + l15) * 4; \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                  \
          const f32x4 v = *reinterpret_cast<const f32x4*>(row + m * 64);                 \
          BUF[t][m] = in ? v : (f32x4){0.f, 0.f, 0.f, 0.f};                              \
        }                                                                                \
      }                                                                                  \
    }                                                                                    \
    if (gen_lane) {                                                                      \
      int64_t pc = (PBASE) + p_local;                                                    \
      if (pc > N - 1) pc = N - 1;                                                        \
      jet = a[pc * KI + f_local];                                                        \
    }                                                                                    \
  }
#define PDEKAN_SKINNY_PASS(BUF, BNEXT, PBASE)                                                          \
  {                                                                                                    \
    if (gen_lane) {                                                                                    \
      float4 e[5];                                                                                     \
      int c0;                                                                                          \
      expand_jets(sK + f_local * kKnotPad, nk, act, jet, e, c0);                                       \
      write_expanded<C1>(tileE, RS, f_local, p_local, e, c0, prev_c0, tileE + (f_local * C1) * RS + ROWS); \
    }                                                                                                  \
    PDEKAN_SKINNY_FETCH(BNEXT, (PBASE) + 4 * PB)                                                       \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                    \
      f32x4 fe[NT];                                                                                    \
      _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                   \
        fe[n] = *reinterpret_cast<const f32x4*>(tileE + (n * 16 + l15) * RS + 16 * t + 4 * q);         \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
        _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                 \
          _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                \
            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fe[n][j], BUF[t][m][j], acc[n][m], 0, 0, 0); \
    }                                                                                                  \
  }
  // wave w takes passes w, w+4, w+8, ... of the block's chunk, two per loop trip (register ping-pong)
  int64_t pb = pbeg + (int64_t)wave * PB;
  if (pb < pend) PDEKAN_SKINNY_FETCH(bA, pb)
  for (; pb < pend; pb += 8 * PB) {
    PDEKAN_SKINNY_PASS(bA, bB, pb)
    if (pb + 4 * PB < pend) PDEKAN_SKINNY_PASS(bB, bA, pb + 4 * PB)
  }
#undef PDEKAN_SKINNY_PASS
#undef PDEKAN_SKINNY_FETCH

  // add the four waves' accumulators: D[column][output], lane holds columns n*16 + 4q + r of output m*16 + l15
  __syncthreads();
  float* red = sE + wave * (NT * 16 * 64);
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(n * 16 + 4 * q + r) * 64 + m * 16 + l15] = acc[n][m][r];
  __syncthreads();
  float* dst0 = part + ((size_t)blockIdx.x * O + o0) * COLS;
  for (int idx = tid; idx < COLS * 64; idx += 256) {
    const int o = idx / COLS, col = idx - o * COLS;
    const float* src = sE + col * 64 + o;
    dst0[(size_t)o * COLS + col] = (src[0] + src[NT * 16 * 64]) + (src[2 * NT * 16 * 64] + src[3 * NT * 16 * 64]);
  }
}

// ==========================================================================
// input-jet gradients: block = 4 waves x 8 points.  Per slab of KIS features
//   gE[row][column] = sum_o gY[row][o] * W[o][column]       (16x16x4 MFMA, A = gY kept in registers)
// then each (point, feature) lane pulls its 4+1 live columns out of LDS and
// applies the chain rule through B', B'', B''' and the activation derivatives.
// The W tile of the next step and the jets the epilogue needs are fetched into
// registers ahead of the MFMA loop.
// ==========================================================================
template <int C1, int KIS, int OT>
__global__ void __launch_bounds__(256, (OT > 1 ? 1 : 2))   // wide layers keep 32*OT gY registers per lane
bwd_input_mfma_kernel(const float4* __restrict__ a, const float4* __restrict__ gy,
                      const float* __restrict__ grid, const float* __restrict__ ginfo, const float* __restrict__ wp,
                      float4* __restrict__ ga, int64_t N, int I, int O, int act) {
  constexpr int C = C1 - 1;
  constexpr int COLS = KIS * C1;
  static_assert(COLS % 16 == 0, "16-wide column tiles");
  constexpr int NT = COLS / 16;
  constexpr int RW = COLS + ((COLS % 32 == 16) ? 0 : 16);   // W slab row stride = 16 mod 32
  constexpr int RT = 66;                                  // gY staging row stride
  constexpr int nk = C1 + 3;
  constexpr int S_FLOATS = 4 * 8 * COLS * 4;              // per-wave [8 points][COLS][4 jets]
  constexpr int W_FLOATS = 64 * RW;
  constexpr int T_FLOATS = 128 * RT;
  constexpr int U_FLOATS = (S_FLOATS > W_FLOATS ? (S_FLOATS > T_FLOATS ? S_FLOATS : T_FLOATS)
                                                : (W_FLOATS > T_FLOATS ? W_FLOATS : T_FLOATS));
  constexpr int WV = (64 * (COLS / 4) + 255) / 256;       // float4 of W tile per thread
  constexpr int NP = (8 * KIS + 63) / 64;                 // (point, feature) pairs per lane
  constexpr int kr = nk + 1;                              // knots + 1/h
  static_assert(KIS * kr <= 2 * 256, "knot rows: at most two floats per thread");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sU = reinterpret_cast<float*>(smem);             // union: gY staging / W tile / gE tiles
  float* sK = sU + U_FLOATS;                              // [KIS][kKnotPad]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, l15 = lane & 15;
  const int64_t PB0 = (int64_t)blockIdx.x * 32;           // first point of the block
  const int64_t PW0 = PB0 + wave * 8;                     // first point of the wave

  // ---- gY of the 32 rows of this wave into registers, 64 outputs at a time ----
  // (macros with literal chunk numbers: every register array below must be
  //  indexed by compile-time constants or it is demoted to scratch memory)
  float afrag[OT][2][16];
#define PDEKAN_STAGE_GY(OC)                                                                        \
  {                                                                                                \
    __syncthreads();                                                                               \
    for (int idx = tid; idx < 32 * 64; idx += 256) {                                               \
      const int p = idx >> 6, o = idx & 63;                                                        \
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                  \
      if (PB0 + p < N) v = gy[(PB0 + p) * O + (OC) * 64 + o];                                      \
      float* dst = sU + (p * 4) * RT + o;                                                          \
      dst[0] = v.x; dst[RT] = v.y; dst[2 * RT] = v.z; dst[3 * RT] = v.w;                           \
    }                                                                                              \
    __syncthreads();                                                                               \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                               \
      _Pragma("unroll") for (int t = 0; t < 16; ++t)                                               \
        afrag[OC][mt][t] = sU[(wave * 32 + mt * 16 + l15) * RT + 4 * t + q];                       \
  }
  PDEKAN_STAGE_GY(0)
  if constexpr (OT > 1) PDEKAN_STAGE_GY(1)
  if constexpr (OT > 2) { PDEKAN_STAGE_GY(2) PDEKAN_STAGE_GY(3) }
#undef PDEKAN_STAGE_GY

  f32x4 wreg[WV];                                         // native vectors: HIP float4 copies pin arrays in scratch
  float kreg0 = 0.f, kreg1 = 0.f;
  f32x4 jete[NP];
  const int k0f = tid / kr, k0k = tid % kr, k1f = (tid + 256) / kr, k1k = (tid + 256) % kr;
#define PDEKAN_FETCH_W(I0, OC)                                                                     \
  _Pragma("unroll") for (int v = 0; v < WV; ++v) {                                                 \
    int idx = tid + v * 256;                                                                       \
    if (idx > 64 * (COLS / 4) - 1) idx = 64 * (COLS / 4) - 1;                                      \
    const int o = idx / (COLS / 4), c4 = idx - o * (COLS / 4);                                     \
    wreg[v] = *reinterpret_cast<const f32x4*>(wp + ((size_t)((OC) * 64 + o) * I + (I0)) * C1 + c4 * 4); \
  }
#define PDEKAN_FETCH_KNOTS(I0)                                                                     \
  {                                                                                                \
    if (tid < KIS * kr) kreg0 = knot_row_value(grid, ginfo, (I0) + k0f, k0k, nk);                  \
    if (tid + 256 < KIS * kr) kreg1 = knot_row_value(grid, ginfo, (I0) + k1f, k1k, nk);            \
  }
  PDEKAN_FETCH_W(0, 0)
  PDEKAN_FETCH_KNOTS(0)

  for (int i0 = 0; i0 < I; i0 += KIS) {
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[mt][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define PDEKAN_CHUNK(OC)                                                                           \
  {                                                                                                \
    __syncthreads(); /* union region free again */                                                 \
    _Pragma("unroll") for (int v = 0; v < WV; ++v) {                                               \
      const int idx = tid + v * 256;                                                               \
      if (idx < 64 * (COLS / 4)) {                                                                 \
        const int o = idx / (COLS / 4), c4 = idx - o * (COLS / 4);                                 \
        *reinterpret_cast<f32x4*>(sU + o * RW + c4 * 4) = wreg[v];                                 \
      }                                                                                            \
    }                                                                                              \
    if ((OC) == 0) {                                                                               \
      if (tid < KIS * kr) sK[k0f * kKnotPad + k0k] = kreg0;                                        \
      if (tid + 256 < KIS * kr) sK[k1f * kKnotPad + k1k] = kreg1;                                  \
    }                                                                                              \
    /* next W tile: next output chunk of this slab, or the first of the next slab */               \
    if ((OC) + 1 < OT) {                                                                           \
      PDEKAN_FETCH_W(i0, (OC) + 1)                                                                 \
    } else if (i0 + KIS < I) {                                                                     \
      PDEKAN_FETCH_W(i0 + KIS, 0)                                                                  \
      PDEKAN_FETCH_KNOTS(i0 + KIS)                                                                 \
    }                                                                                              \
    if ((OC) == OT - 1) {                                                                          \
      _Pragma("unroll") for (int v = 0; v < NP; ++v) {                                             \
        const int pair = lane + 64 * v;                                                            \
        int64_t pc = PW0 + (pair & 7);                                                             \
        if (pc > N - 1) pc = N - 1;                                                                \
        const int f = (pair >> 3) < KIS ? (pair >> 3) : 0;                                         \
        jete[v] = *reinterpret_cast<const f32x4*>(a + pc * I + i0 + f);                            \
      }                                                                                            \
    }                                                                                              \
    __syncthreads();                                                                               \
    /* W operands of step t+1 are read while the MFMAs of step t run */                           \
    float fw[2][NT];                                                                               \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) fw[0][n] = sU[q * RW + n * 16 + l15];           \
    _Pragma("unroll") for (int t = 0; t < 16; ++t) { /* full: afrag needs constant indices */     \
      if (t + 1 < 16) {                                                                            \
        _Pragma("unroll") for (int n = 0; n < NT; ++n)                                             \
          fw[(t + 1) & 1][n] = sU[(4 * (t + 1) + q) * RW + n * 16 + l15];                          \
      }                                                                                            \
      __builtin_amdgcn_sched_barrier(0); /* keep the reads ahead of the MFMAs */                   \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                             \
        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[OC][0][t], fw[t & 1][n], acc[0][n], 0, 0, 0); \
        acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[OC][1][t], fw[t & 1][n], acc[1][n], 0, 0, 0); \
      }                                                                                            \
    }                                                                                              \
  }
    PDEKAN_CHUNK(0)
    if constexpr (OT > 1) PDEKAN_CHUNK(1)
    if constexpr (OT > 2) { PDEKAN_CHUNK(2) PDEKAN_CHUNK(3) }

    __syncthreads();                                      // every wave is done with the W tile
    float4* sS = reinterpret_cast<float4*>(sU) + wave * 8 * COLS;     // [8][COLS] float4
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        sS[(mt * 4 + q) * COLS + n * 16 + l15] = make_float4(acc[mt][n][0], acc[mt][n][1], acc[mt][n][2], acc[mt][n][3]);
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NP; ++v) {
      const int pair = lane + 64 * v;
      const int p_local = pair & 7, f_local = pair >> 3;
      const int64_t p = PW0 + p_local;
      if (f_local < KIS && p < N) {
        const float4 jet = make_float4(jete[v][0], jete[v][1], jete[v][2], jete[v][3]);
        Basis bs;
        eval_basis_row<true>(sK + f_local * kKnotPad, nk, jet.x, bs);
        const Act av = eval_act(act, jet.x);
        const float4* col = sS + p_local * COLS + f_local * C1;
        const float s2 = jet.y * jet.y;
        float ox = 0.f, oy = 0.f, oz = 0.f, ow = 0.f;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
          float e1, e2, e3;
          float4 g;
          if (m < 4) {
            const int c = bs.c0 + m;
            const bool ok = (c >= 0) && (c < C);
            g = ok ? col[ok ? c : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
            e1 = bs.d1[m & 3]; e2 = bs.d2[m & 3]; e3 = bs.d3[m & 3];
          } else {
            g = col[C];
            e1 = av.f1; e2 = av.f2; e3 = av.f3;
          }
          ox += g.x * e1 + (g.y * jet.y + g.z * jet.z) * e2 + g.w * (s2 * e3 + jet.w * e2);
          oy += g.y * e1 + 2.0f * jet.y * g.w * e2;
          oz += g.z * e1;
          ow += g.w * e1;
        }
        ga[p * I + i0 + f_local] = make_float4(ox, oy, oz, ow);
      }
    }
  }
}

#undef PDEKAN_CHUNK
#undef PDEKAN_FETCH_W
#undef PDEKAN_FETCH_KNOTS

// ==========================================================================
// host side
// ==========================================================================
struct MfmaCfg { int ki_fwd, kiw, kis; };

static bool cfg_for(int G, MfmaCfg& c) {
  if (G == 10) { c = {4, 8, 8}; return true; }
  if (G == 20) { c = {2, 4, 4}; return true; }
  if (G == 5) { c = {4, 16, 16}; return true; }
  return false;
}

bool mfma_eligible(int D, int D2, const LayerView& L) {
  if (D != 2 || D2 != 1) return false;
  const char* off = getenv("PDEKAN_DISABLE_MFMA");      // A/B switch for tests and profiling
  if (off && off[0] == '1') return false;
  MfmaCfg c;
  if (!cfg_for(L.G, c)) return false;
  if (L.O % 64 != 0 || L.O > 256 || L.O == 192) return false;
  if (L.I < 32 || L.I % (4 * c.kiw) != 0) return false;
  return true;
}

// Chunks of points per weight-gradient launch: enough blocks to fill the chip, and short enough that a chunk's
// running fp32 sums stay accurate (a 2^19-point chunk loses 3e-5 against an fp64 sum, a 4096-point one 2e-6; the
// chunks themselves are added in 64 interleaved running sums by sum_partials_kernel), within 4 GiB of partial sums.
int mfma_weight_chunks(int64_t N, const LayerView& L) {
  MfmaCfg c;
  cfg_for(L.G, c);
  const int per_chunk = (L.I / (4 * c.kiw)) * (L.O / 64);
  int64_t want = (512 + per_chunk - 1) / per_chunk;
  int64_t chunk_pts = 4096;
  if (const char* e = getenv("PDEKAN_WEIGHT_CHUNK_POINTS")) {          // experiments: 512 .. 2^20 points per chunk
    const long v = atol(e);
    if (v >= 512 && v <= (1 << 20)) chunk_pts = v;
  }
  const int64_t for_accuracy = (N + chunk_pts - 1) / chunk_pts;
  if (want < for_accuracy) want = for_accuracy;
  const int64_t fits = ((int64_t)4 << 30) / ((int64_t)L.packed_floats() * 4);
  if (want > fits) want = fits;
  const int64_t cap = (N + 511) / 512;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

template <typename Kern>
static void allow_big_lds(Kern k, size_t bytes) {
  if (bytes > 48 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(k), bytes);
}

template <int C1, int KI>
static void fwd_launch(const LayerView& L, const float* a, const float* wt, float* y, int64_t N, hipStream_t st) {
  constexpr int KK = KI * C1;
  const size_t lds = (size_t)(5 * KK * 64 + 2 * KI * kKnotPad + 256 * 4) * 4;
  auto k = fwd_mfma_kernel<C1, KI>;
  allow_big_lds(k, lds);
  dim3 g((unsigned)((N + 63) / 64), L.O / 64);
  k<<<g, 256, lds, st>>>(reinterpret_cast<const float4*>(a), L.grid, L.ginfo, wt, reinterpret_cast<float4*>(y), N, L.I,
                         L.O, L.act);
}

void launch_fwd_mfma(const LayerView& L, const float* a, const float* wt, float* y, int64_t N, hipStream_t st) {
  if (L.G == 10) fwd_launch<14, 4>(L, a, wt, y, N, st);
  else if (L.G == 20) fwd_launch<24, 2>(L, a, wt, y, N, st);
  else fwd_launch<9, 4>(L, a, wt, y, N, st);
}

// split-bf16 weight gradients: grid 10 (8 features x 14 columns = 7 column tiles per wave) and grid 20 (4 features
// x 24 columns = 6 column tiles).  The kernel walks whole pairs of
// 8-point steps; when N is not a multiple of 16 the last N % 16 points go to the fp32 kernel as a chunk of their own
// (so at least two chunks are needed).
bool bf16x3_weight_eligible(int D, int D2, const LayerView& L, int64_t N) {
  if (!mfma_eligible(D, D2, L) || (L.G != 10 && L.G != 20) || N < 16 || !L.ginfo) return false;
  if (N % 16 != 0 && mfma_weight_chunks(N, L) < 2) return false;
  const char* off = getenv("PDEKAN_DISABLE_BF16X3");    // A/B switch for tests and profiling
  return !(off && off[0] == '1');
}

template <int C1, int KIW, int PB>
static void bww_launch(const LayerView& L, const float* a, const float* gy, float* part, int nchunks, int64_t N,
                       hipStream_t st, bool with_bf16x3 = false, const unsigned* act_in = nullptr,
                       const unsigned* cot_in = nullptr, bool skip_layer = false) {
  constexpr int COLS = KIW * C1, ROWS = PB * 4, RS = ROWS + 4;
  const size_t lds = (size_t)(4 * COLS * RS + ROWS * 80 + 4 * KIW * kKnotPad) * 4;
  auto k = bwd_weight_mfma_kernel<C1, KIW, PB>;
  allow_big_lds(k, lds);
  const int nx = L.I / (4 * KIW), ny = L.O / 64;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* g4 = reinterpret_cast<const float4*>(gy);
  if (!with_bf16x3) {
    int64_t chunk = (N + nchunks - 1) / nchunks;
    chunk = (chunk + PB - 1) / PB * PB;
    k<<<(unsigned)(nx * ny * nchunks), 256, lds, st>>>(a4, g4, L.grid, L.ginfo, part, N, L.I, L.O, L.act, chunk, nx, ny,
                                                       nchunks, skip_layer ? 2 : 0);
    return;
  }
  if constexpr (C1 % 2 == 0 && (KIW == 8 || KIW == 4) && PB == 8) {
    // points [0, N16) in nmain chunks: every block of 4 * KIW features goes to exactly one of the two kernels
    const int64_t N16 = N / 16 * 16, tail = N - N16;
    const int nmain = tail ? nchunks - 1 : nchunks;
    int64_t chunk = (N16 + nmain - 1) / nmain;
    chunk = (chunk + 15) / 16 * 16;
    k<<<(unsigned)(nx * ny * nmain), 256, lds, st>>>(a4, g4, L.grid, L.ginfo, part, N16, L.I, L.O, L.act, chunk, nx, ny,
                                                     nmain, 1);
    if (act_in && cot_in) {                              // fp16 two-piece kernels, same block decode and selection
      launch_bwd_weight_f16_blocks(L, a, gy, part, N16, chunk, nx, ny, nmain, act_in, cot_in, st);
    } else {
    // both variants on the same grid: a block runs in the one that matches its features' knot rows
    auto kb = (L.act == kActSiLU) ? bwd_weight_bf16x3_kernel<C1, KIW, kActSiLU, false>
                                  : bwd_weight_bf16x3_kernel<C1, KIW, kActIdentity, false>;
    auto kg = (L.act == kActSiLU) ? bwd_weight_bf16x3_kernel<C1, KIW, kActSiLU, true>
                                  : bwd_weight_bf16x3_kernel<C1, KIW, kActIdentity, true>;
    const size_t lds_b = (size_t)4 * (3 * KIW * C1 * kBf16Rw + KIW * kKnotPad) * 4;
    const size_t lds_g = (size_t)4 * (3 * KIW * C1 * kBf16Rw + KIW * kKnotPadGen) * 4;
    allow_big_lds(kb, lds_b);
    allow_big_lds(kg, lds_g);
    kb<<<(unsigned)(nx * ny * nmain), 256, lds_b, st>>>(a4, g4, L.grid, L.ginfo, part, N16, L.I, L.O, chunk, nx, ny, nmain);
    kg<<<(unsigned)(nx * ny * nmain), 256, lds_g, st>>>(a4, g4, L.grid, L.ginfo, part, N16, L.I, L.O, chunk, nx, ny, nmain);
    }
    if (tail)                                            // the last chunk slot: the points past N16, fp32 kernel, all blocks
      k<<<(unsigned)(nx * ny), 256, lds, st>>>(a4 + (size_t)N16 * L.I, g4 + (size_t)N16 * L.O, L.grid, L.ginfo,
                                               part + (size_t)nmain * L.packed_floats(), tail, L.I, L.O, L.act, 16, nx, ny,
                                               1, 0);
  }
}

void launch_bwd_weight_mfma(const LayerView& L, const float* a, const float* gy, float* part, int nchunks, int64_t N,
                            hipStream_t st, bool with_bf16x3, const unsigned* act_in, const unsigned* cot_in,
                            bool skip_layer) {
  if (L.G == 10) bww_launch<14, 8, 8>(L, a, gy, part, nchunks, N, st, with_bf16x3, act_in, cot_in, skip_layer);
  else if (L.G == 20) bww_launch<24, 4, 8>(L, a, gy, part, nchunks, N, st, with_bf16x3, act_in, cot_in, skip_layer);
  else bww_launch<9, 16, 4>(L, a, gy, part, nchunks, N, st);
}

bool mfma_skinny_weight_eligible(int D, int D2, const LayerView& L) {
  if (D != 2 || D2 != 1) return false;
  const char* off = getenv("PDEKAN_DISABLE_MFMA");
  if (off && off[0] == '1') return false;
  MfmaCfg c;
  if (!cfg_for(L.G, c)) return false;
  return L.I == 2 && L.O % 64 == 0 && L.O <= 256;
}

int mfma_skinny_weight_chunks(int64_t N, const LayerView& L) {
  int64_t want = 768 / (L.O / 64);                      // three blocks per CU in flight (the register budget's limit)
  const int64_t for_accuracy = (N + 16383) / 16384;     // a chunk's 4 waves keep running fp32 sums over a quarter of it each
  if (want < for_accuracy) want = for_accuracy;
  const int64_t cap = (N + 255) / 256;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

template <int C1>
static void bww_skinny_launch(const LayerView& L, const float* a, const float* gy, float* part, int nchunks, int64_t N,
                              hipStream_t st) {
  constexpr int KI = 2, NT = (KI * C1 + 15) / 16;
  const size_t lds = (size_t)(4 * NT * 16 * 64 + KI * kKnotPad) * 4;
  int64_t chunk = (N + nchunks - 1) / nchunks;
  chunk = (chunk + 31) / 32 * 32;
  auto k = bwd_weight_mfma_skinny_kernel<C1, KI>;
  allow_big_lds(k, lds);
  dim3 g(nchunks, L.O / 64);
  k<<<g, 256, lds, st>>>(reinterpret_cast<const float4*>(a), gy, L.grid, L.ginfo, part, N, L.O, L.act, chunk);
}

void launch_bwd_weight_mfma_skinny(const LayerView& L, const float* a, const float* gy, float* part, int nchunks,
                                   int64_t N, hipStream_t st) {
  if (L.G == 10) bww_skinny_launch<14>(L, a, gy, part, nchunks, N, st);
  else if (L.G == 20) bww_skinny_launch<24>(L, a, gy, part, nchunks, N, st);
  else bww_skinny_launch<9>(L, a, gy, part, nchunks, N, st);
}

template <int C1, int KIS, int OT>
static void bwi_launch(const LayerView& L, const float* a, const float* gy, const float* wp, float* ga, int64_t N,
                       hipStream_t st) {
  constexpr int COLS = KIS * C1;
  constexpr int RW = COLS + ((COLS % 32 == 16) ? 0 : 16);
  constexpr int S = 4 * 8 * COLS * 4, W = 64 * RW, T = 128 * 66;
  constexpr int U = (S > W ? (S > T ? S : T) : (W > T ? W : T));
  const size_t lds = (size_t)(U + KIS * kKnotPad) * 4;
  auto k = bwd_input_mfma_kernel<C1, KIS, OT>;
  allow_big_lds(k, lds);
  k<<<(unsigned)((N + 31) / 32), 256, lds, st>>>(reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(gy),
                                                 L.grid, L.ginfo, wp, reinterpret_cast<float4*>(ga), N, L.I, L.O, L.act);
}

template <int C1, int KIS>
static void bwi_dispatch(const LayerView& L, const float* a, const float* gy, const float* wp, float* ga, int64_t N,
                         hipStream_t st) {
  if (L.O == 64) bwi_launch<C1, KIS, 1>(L, a, gy, wp, ga, N, st);
  else if (L.O == 128) bwi_launch<C1, KIS, 2>(L, a, gy, wp, ga, N, st);
  else bwi_launch<C1, KIS, 4>(L, a, gy, wp, ga, N, st);
}

void launch_bwd_input_mfma(const LayerView& L, const float* a, const float* gy, const float* wp, float* ga, int64_t N,
                           hipStream_t st) {
  if (L.G == 10) bwi_dispatch<14, 8>(L, a, gy, wp, ga, N, st);
  else if (L.G == 20) bwi_dispatch<24, 4>(L, a, gy, wp, ga, N, st);
  else bwi_dispatch<9, 16>(L, a, gy, wp, ga, N, st);
}

}  // namespace pdekan
