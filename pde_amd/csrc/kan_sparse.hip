// Sparse-support VALU kernels for hidden KAN layers on gfx950, J = 4 jets
// (value, d/dS, d/dt, d2/dS2): forward and input-jet gradients.
//
// Of the C+1 = G+4 columns a feature contributes to the contraction
// (efficient_kan/kan.py:153-166), only the 4 cubic bases whose support holds x
// and the base-activation column are non-zero for a given point.  A dense GEMM
// multiplies the other G-1 columns by zero; on gfx950 the fp32 MFMA runs at the
// fp32 VECTOR rate, so skipping them on the VALU is a net win: 5/(G+4) of the
// multiply-adds (2.8x fewer at G = 10).
//
// Lane = (point, group of 32 outputs).  A lane evaluates the basis of its
// (point, feature) once, finds the interval index s of x and then needs, per
// output, the 4 consecutive spline weights w[o][i][s-3 .. s]: these come from a
// "window table" in LDS laid out [output][slot s][4 weights] - 16 slots x 16 B =
// one 256-byte bank row per output, so the 16 lanes of a ds_read_b128 group hit
// either the same slot (broadcast) or different banks: conflict-free whatever
// the data.  The table of feature i+1 is staged (global -> registers -> LDS,
// double-buffered) while feature i is being consumed.
//
// A block takes 64 of a layer's outputs.  Wider layers go 64 at a time: the
// forward grid has one block row per 64 outputs, the input-gradient kernel
// walks the outputs 64 at a time and adds each pass into the gradients it
// stored in the pass before (template parameter MG).
//
// Layers the window tables do not fit (more than 16 knot intervals: grid 20; 192
// or 256 outputs) keep the weights as raw rows [output][3 zeros, C weights,
// zeros] (32 floats): the 4 live weights are floats s .. s+3 of the row, read
// at 4-byte alignment (two ds_read2_b32: half the LDS rate of the aligned
// read, a quarter of the table size).
//
//   forward:      y[p][o][:]     += sum_m w[o][i][s-3+m] * E_m[:] + w_b[o][i] * E_b[:]
//   input grads:  G_m[:]          = sum_o w[o][i][s-3+m] * gy[p][o][:]   (+ base), then the
//                 chain rule through B', B'', B''' exactly as in the streaming kernel
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kan_math.h"
#include "kan_launch.h"
#include "kan_wave.h"

namespace pdekan {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSlots = 16;          // interval slots per (output, feature) row: needs G + 6 <= 16
constexpr int kKnotPadS = 28;

// wp[O][I][C1] -> win[I][O][16][4] (window of 4 spline weights per interval slot), wbt[I][O]
__global__ void pack_window_kernel(const float* __restrict__ wp, float* __restrict__ win,
                                   float* __restrict__ wbt, int I, int O, int C) {
  const int C1 = C + 1;
  const int total = I * O * kSlots;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int slot = t % kSlots;
    const int o = (t / kSlots) % O;
    const int i = t / (kSlots * O);
    const float* w = wp + ((size_t)o * I + i) * C1;
    f32x4 v;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int c = slot - 3 + m;
      v[m] = (c >= 0 && c < C) ? w[c] : 0.0f;
    }
    *reinterpret_cast<f32x4*>(win + (size_t)t * 4) = v;
    if (slot == 0) wbt[(size_t)i * O + o] = w[C];
  }
}

// wp[O][I][C1] -> raw[I][O][32] = {0, 0, 0, the C spline weights, 0 ...}: the 4 weights live on knot interval s are
// the floats s .. s+3 of the row (kRawRow >= G + 10), wbt[I][O]
constexpr int kRawRow = 32;
__global__ void pack_raw_rows_kernel(const float* __restrict__ wp, float* __restrict__ raw,
                                     float* __restrict__ wbt, int I, int O, int C) {
  const int C1 = C + 1;
  const int total = I * O * kRawRow;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int col = t % kRawRow;
    const int o = (t / kRawRow) % O;
    const int i = t / (kRawRow * O);
    const float* w = wp + ((size_t)o * I + i) * C1;
    const int c = col - 3;
    raw[t] = (c >= 0 && c < C) ? w[c] : 0.0f;
    if (col == 0) wbt[(size_t)i * O + o] = w[C];
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc += v * w.x  /  acc += v * w.y  on a register pair (two fp32 lanes per instruction).
// Written out because the compiler materialises some of these broadcasts with an
// extra v_mov per use instead of the operand-select bits.
__device__ __forceinline__ void pkfma_x(f32x2& acc, f32x2 v, f32x2 w) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(v), "v"(w));
}
__device__ __forceinline__ void pkfma_y(f32x2& acc, f32x2 v, f32x2 w) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(v), "v"(w));
}
__device__ __forceinline__ f32x2 lo2(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }

// Block barrier that also retires the LDS-DMA loads this wave has in flight: they
// count on the vector-memory counter, which a plain __syncthreads() does not
// always wait for, so the wait is spelled out.
__device__ __forceinline__ void barrier_after_dma() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Basis and activation of one (point, feature) for the sparse kernels.
struct Live {
  f32x4 b, d1, d2, d3;   // the 4 live bases and derivatives
  Act act;
  int slot;              // interval index s, clamped to [0, kSlots)
};

template <bool kNeedD3, int kMaxSlot = kSlots - 1>
__device__ __forceinline__ Live eval_live(const float* knots, int nk, int act_kind, float x) {
  Basis bs;
  eval_basis_row<kNeedD3>(knots, nk, x, bs);
  Live L;
#pragma unroll
  for (int m = 0; m < 4; ++m) { L.b[m] = bs.b[m]; L.d1[m] = bs.d1[m]; L.d2[m] = bs.d2[m]; L.d3[m] = bs.d3[m]; }
  L.act = eval_act(act_kind, x);
  int s = bs.c0 + 3;
  s = s < 0 ? 0 : (s > kMaxSlot ? kMaxSlot : s);
  L.slot = s;
  return L;
}

// --------------------------------------------------------------------------
// Shared staging of one feature's tables (O*64 floats of windows + O base
// weights): a straight copy, so the four waves move it with LDS-DMA loads, one
// KiB per instruction, no registers involved.  Retired by the next barrier_after_dma().
// --------------------------------------------------------------------------
// RS = floats per output in the table: 64 (window table, 16 slots x 4) or kRawRow (raw rows).  The block takes the
// O outputs from o0 of a layer with OT outputs (tables are [I][OT][RS], base weights [I][OT]).
template <int O, int RS = 64>
__device__ __forceinline__ void table_stage(float* sT, float* sB, const float* win, const float* wbt, int i, int wave,
                                            int lane, int OT = O, int o0 = 0) {
  static_assert((O * RS) % 1024 == 0, "each wave copies whole 1 KiB pieces");
  const float* src = win + ((size_t)i * OT + o0) * RS + wave * (O * RS / 4);
  float* dst = sT + wave * (O * RS / 4);
#pragma unroll
  for (int k = 0; k < O * RS / 1024; ++k)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(dst + k * 256), 16, 0, 0);
  if (wave < O / 64)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbt + (size_t)i * OT + o0 + wave * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(sB + wave * 64), 4, 0, 0);
}

// the 4 live spline weights of one output: one 16-byte read of its window slot, or 4 floats from its raw row at
// any 4-byte offset (two ds_read2_b32)
template <int RS>
__device__ __forceinline__ f32x4 live_weights(const float* row) {
  if constexpr (RS == 64) return *reinterpret_cast<const f32x4*>(row);
  else return (f32x4){row[0], row[1], row[2], row[3]};
}

// --------------------------------------------------------------------------
// Input jets a[N][I][4] are consumed one feature at a time by lanes that each own
// a point: read directly that is a 16-byte load per lane at a 16*I-byte stride.
// Instead a wave stages kFS = 4 features of its PW points at a time: 4 lanes
// fetch the 64 contiguous bytes of one point, the wave parks them in LDS
// ([point][feature][4]) and every lane then picks its point's jet from there.
// --------------------------------------------------------------------------
constexpr int kFS = 4;

// The copy is an LDS-DMA load (no VGPR destination): lane l of chunk k lands at
// base + (64*k + l)*16 bytes, which is exactly [point][feature][4] for point =
// (64*k + l)/4.  The next barrier_after_dma() retires it.
//
// WaveRows holds the wave's view of a [N][I][4] tensor: a wave-uniform base and
// the per-lane element offsets of its chunks (rows past the end of the batch are
// clamped onto the last live row for loads and masked for stores).
template <int PW, int FS = kFS>
struct WaveRows {
  static constexpr int NV = (PW * FS) / 64;
  static_assert((PW * FS) % 64 == 0, "whole 16-byte chunks per lane");
  int64_t base;      // first point of the wave (clamped into the batch), wave-uniform
  int nlive;         // live points of the wave, wave-uniform
  int off[NV];       // (row * I + feature-in-stage) * 4
};

template <int PW, int FS = kFS>
__device__ __forceinline__ WaveRows<PW, FS> wave_rows(int64_t P0, int64_t N, int I, int lane) {
  WaveRows<PW, FS> r;
  int64_t left = N - P0;
  r.nlive = left <= 0 ? 0 : (left < PW ? (int)left : PW);
  r.base = r.nlive > 0 ? P0 : N - 1;
#pragma unroll
  for (int k = 0; k < WaveRows<PW, FS>::NV; ++k) {
    const int c = lane + 64 * k;
    const int row = c / FS;
    const int rc = row < r.nlive ? row : (r.nlive > 0 ? r.nlive - 1 : 0);
    r.off[k] = (rc * I + (c % FS)) * 4;
  }
  return r;
}

template <int PW, int FS = kFS>
__device__ __forceinline__ void jets_stage(float* sJ, const float* a, const WaveRows<PW, FS>& r, int I, int i0) {
  const float* src = a + (r.base * I + i0) * 4;
#pragma unroll
  for (int k = 0; k < WaveRows<PW, FS>::NV; ++k)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r.off[k]),
                                     (__attribute__((address_space(3))) void*)(sJ + 64 * k * 4), 16, 0, 0);
}

// One feature of one point into the lane's 32 outputs x 4 jets: the expanded jets of the 4 live splines and of the
// base column times the outputs' weights.  tb = the lane's first output's window slot (or raw-row position) in the
// feature's table, bb = its base weights.
template <int RS>
__device__ __forceinline__ void feature_into_outputs(f32x2 (&accl)[32], f32x2 (&acch)[32], const Live& L, const f32x4& cur,
                                                     const float* tb, const float* bb) {
  const float s2 = cur[1] * cur[1];
  f32x2 el[5], eh[5];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    el[m] = (f32x2){L.b[m], L.d1[m] * cur[1]};
    eh[m] = (f32x2){L.d1[m] * cur[2], L.d2[m] * s2 + L.d1[m] * cur[3]};
  }
  el[4] = (f32x2){L.act.f0, L.act.f1 * cur[1]};
  eh[4] = (f32x2){L.act.f1 * cur[2], L.act.f2 * s2 + L.act.f1 * cur[3]};
  // outputs in groups of 4; the weights of group q+2 are read from LDS while group q is consumed
  f32x4 wq[3][4], bq[3];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int k = 0; k < 4; ++k) wq[r][k] = live_weights<RS>(tb + (r * 4 + k) * RS);
    bq[r] = *reinterpret_cast<const f32x4*>(bb + r * 4);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int cq = q % 3, nxt = (q + 2) % 3;
    if (q + 2 < 8) {
#pragma unroll
      for (int k = 0; k < 4; ++k) wq[nxt][k] = live_weights<RS>(tb + ((q + 2) * 4 + k) * RS);
      bq[nxt] = *reinterpret_cast<const f32x4*>(bb + (q + 2) * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x2 w01 = lo2(wq[cq][k]), w23 = hi2(wq[cq][k]);
      const f32x2 wb = (k < 2) ? lo2(bq[cq]) : hi2(bq[cq]);
      f32x2& tl = accl[q * 4 + k];
      f32x2& th = acch[q * 4 + k];
      pkfma_x(tl, el[0], w01); pkfma_x(th, eh[0], w01);
      pkfma_y(tl, el[1], w01); pkfma_y(th, eh[1], w01);
      pkfma_x(tl, el[2], w23); pkfma_x(th, eh[2], w23);
      pkfma_y(tl, el[3], w23); pkfma_y(th, eh[3], w23);
      if (k & 1) { pkfma_y(tl, el[4], wb); pkfma_y(th, eh[4], wb); }
      else       { pkfma_x(tl, el[4], wb); pkfma_x(th, eh[4], wb); }
    }
  }
}

// A lane holds 32 outputs x 4 jets of one point (512 contiguous bytes in y).  Stored straight from registers every
// instruction would touch 64 different lines 16 bytes at a time, so the wave's tile goes through its own LDS area
// (64 lines of kParkLine floats) 8 outputs at a time and leaves as whole 128-byte lines (8 lanes per line).
constexpr int kParkLine = 36;                            // floats per parked line: 32 + 4 pad (bank spread)
// FULL: every point of the tile is live; the stores are then unconditional, which keeps the epilogue one basic block
// (with branches around the stores the compiler can no longer count them and waits for ALL of them wherever the
// code behind needs an earlier load).
template <int OG, bool FULL = false>
__device__ __forceinline__ void store_outputs_through_lds(float* park, const f32x2 (&accl)[32], const f32x2 (&acch)[32],
                                                          float* __restrict__ ytile, int nlive, int OT, int o0,
                                                          int lane, int g, int pl) {   // ytile = row of the tile's first point
  static_assert(8 % OG == 0, "a pass of 8 lines is whole points");
  const int line_w = pl * OG + g;                        // the line this lane fills: (point, output group)
  // the line this lane stores in pass r is (lane >> 3) + 8r: point row0 + (8/OG) r, always the same output group;
  // everything that changes with r or k is wave-uniform and goes into the scalar part of the address
  const int row0 = (lane >> 3) / OG;
  const int lane_off = (row0 * OT + o0 + ((lane >> 3) % OG) * 32 + (lane & 7)) * 4;
  const float* rd = park + (lane >> 3) * kParkLine + (lane & 7) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int o = 8 * k + j;
      *reinterpret_cast<f32x4*>(park + line_w * kParkLine + j * 4) = (f32x4){accl[o][0], accl[o][1], acch[o][0], acch[o][1]};
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(rd + 8 * r * kParkLine);
      float* yr = ytile + ((size_t)(r * (8 / OG)) * OT + 8 * k) * 4;
      if (FULL || row0 + r * (8 / OG) < nlive) *reinterpret_cast<f32x4*>(yr + lane_off) = v;
    }
  }
}

// live points of a tile of PW points that starts at P0
__device__ __forceinline__ int tile_live(int64_t P0, int64_t N, int PW) {
  const int64_t left = N - P0;
  return left <= 0 ? 0 : (left < PW ? (int)left : PW);
}

// ==========================================================================
// forward
// ==========================================================================
// MG (several groups of O outputs per layer): blockIdx.y picks the block's O of the layer's OTr outputs
template <int O, int FS, int RS = 64, bool MG = (RS != 64)>
__global__ void __launch_bounds__(256, 2)
fwd_sparse_kernel(const float* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
                  const float* __restrict__ win,
                  const float* __restrict__ wbt, float* __restrict__ y, int64_t N, int I, int G, int act, int OTr,
                  int skip_near_uniform, unsigned* __restrict__ amax_out) {
  if (skip_near_uniform && all_rows_near_uniform(ginfo, I)) return;   // the fp16 matrix kernel beside this one takes the layer
  constexpr int OG = O / 32;              // output groups = lanes per point
  constexpr int PW = 64 / OG;             // points per wave
  constexpr int kMaxSlot = (RS == 64) ? kSlots - 1 : kRawRow - 4;
  const int OT = MG ? OTr : O;
  const int o0 = MG ? (int)blockIdx.y * O : 0;
  const int nk = G + 7;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = reinterpret_cast<float*>(smem);            // [2][O*RS]
  float* sB = sT + 2 * O * RS;                           // [2][O]
  float* sJ = sB + 2 * O;                                // [4 waves][2][PW][FS][4] staged input jets
  float* sK = sJ + 4 * 2 * PW * FS * 4;                 // [I][kKnotPadS]

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int g = lane / PW, pl = lane % PW;
  const int64_t P0 = ((int64_t)blockIdx.x * 4 + wave) * PW;     // first point of the wave
  int64_t p = P0 + pl;
  const bool live_pt = p < N;
  if (!live_pt) p = N - 1;
  float* sJw = sJ + wave * 2 * PW * FS * 4;
  const WaveRows<PW, FS> wr = wave_rows<PW, FS>(P0, N, I, lane);

  for (int t = tid; t < I * (nk + 1); t += 256) {        // knot rows + the feature's 1/h (0 = not equally spaced)
    const int f = t / (nk + 1), k = t - f * (nk + 1);
    sK[f * kKnotPadS + k] = (k < nk) ? grid[(size_t)f * nk + k] : (ginfo ? ginfo[f] : 0.0f);
  }
  table_stage<O, RS>(sT, sB, win, wbt, 0, wave, lane, OT, o0);
  jets_stage<PW, FS>(sJw, a, wr, I, 0);

  f32x2 accl[32], acch[32];               // jets (0,1) and (2,3) of the lane's 32 outputs
#pragma unroll
  for (int o = 0; o < 32; ++o) { accl[o] = (f32x2){0.f, 0.f}; acch[o] = (f32x2){0.f, 0.f}; }

  // The basis of feature i+1 is evaluated at the END of iteration i, before the barrier: the chain
  // jet -> interval -> basis -> slot then overlaps the other waves' way to the barrier, and the first
  // weight reads go out right after it.
  barrier_after_dma();                     // knots, table 0, jets of features 0..3
  f32x4 cur = *reinterpret_cast<const f32x4*>(sJw + (pl * FS) * 4);
  Live L = eval_live<false, kMaxSlot>(sK, nk, act, cur[0]);

  for (int i = 0; i < I; ++i) {
    const int buf = i & 1;
    if (i + 1 < I)
      table_stage<O, RS>(sT + (buf ^ 1) * O * RS, sB + (buf ^ 1) * O, win, wbt, i + 1, wave, lane, OT, o0);
    if ((i % FS) == 0 && i + FS < I)     // features i+4..i+7 into the other staging buffer
      jets_stage<PW, FS>(sJw + (((i / FS) + 1) & 1) * PW * FS * 4, a, wr, I, i + FS);
    feature_into_outputs<RS>(accl, acch, L, cur, sT + buf * O * RS + (g * 32) * RS + L.slot * (RS == 64 ? 4 : 1),
                             sB + buf * O + g * 32);
    if (i + 1 < I) {
      const int n = i + 1;
      cur = *reinterpret_cast<const f32x4*>(sJw + (((n / FS) & 1) * PW * FS + pl * FS + (n % FS)) * 4);
      L = eval_live<false, kMaxSlot>(sK + n * kKnotPadS, nk, act, cur[0]);
      barrier_after_dma();                 // table i+1 landed everywhere; table i no longer read
    }
  }
  if (amax_out) {                          // per-jet maxima of the outputs, for the scales of an fp16 matrix layer above
    f32x2 ml = {0.f, 0.f}, mh = {0.f, 0.f};
    if (live_pt) {
#pragma unroll
      for (int o = 0; o < 32; ++o) {
        ml[0] = fmaxf(ml[0], fabsf(accl[o][0])); ml[1] = fmaxf(ml[1], fabsf(accl[o][1]));
        mh[0] = fmaxf(mh[0], fabsf(acch[o][0])); mh[1] = fmaxf(mh[1], fabsf(acch[o][1]));
      }
    }
    publish_jet_absmax(ml[0], ml[1], mh[0], mh[1], amax_out);
  }
  // Epilogue through LDS (store_outputs_through_lds): each wave's area spans sT, sB and sJ (raw rows: also the head
  // of the knot rows, dead by now; the launcher sizes the allocation for it)
  __syncthreads();                                       // every wave is done with the tables
  static_assert(RS != 64 || 4 * 64 * kParkLine <= 2 * O * 64 + 2 * O + 4 * 2 * PW * FS * 4, "parking area fits the staging buffers");
  store_outputs_through_lds<OG>(sT + wave * (64 * kParkLine), accl, acch, y + P0 * OT * 4, wr.nlive, OT, o0, lane, g, pl);
}

// ==========================================================================
// forward of a layer with ONE or TWO inputs (the first layer of a PINN on (S, t)).  With so few features there is
// little arithmetic per output byte and the mapping above (a lane keeps 32 outputs x 4 jets of a point) spends its
// time moving results through LDS to get whole lines out, two waves per SIMD.  Here
//   * one block of 16 waves per compute unit stages the tables of ALL the layer's features once and stays: each
//     wave then walks over batches of 32 points on its own (no block barrier after the prologue);
//   * a batch is evaluated once: lane = one (point, feature), which leaves the 5 expanded jets of that pair and its
//     interval slot in the wave's LDS area;
//   * the products run 8 points at a time with lane = (point, output mod 8): the lane's 8 outputs oc, oc+8, ... are
//     32 accumulator registers, the pair's jets are broadcast reads, and for every j the 8 lanes of a point hold
//     outputs 8j..8j+7, so each store instruction writes 8 whole 128-byte lines straight from registers.
// Table rows are 68 floats apart in LDS (not 64): the 8 output rows a 16-byte read touches land on different banks.
// With its stores switched off the kernel runs in half the time: what bounds it is writing N x O x 16 bytes.
// ==========================================================================
constexpr int kFewInputsMax = 2;
constexpr int kFewRow = 68;                  // floats per (feature, output) row of the staged window tables
constexpr int kFewPair = 20;                 // floats per evaluated (point, feature): 5 columns x 4 jets
constexpr int kFewBatch = 32;                // points per wave and batch
constexpr int kFewWaves = 16;

__device__ __forceinline__ void absmax3_into(float& m, float a, float b) {   // m = max(m, |a|, |b|)
  asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
}

// the 4 steps of 8 points of one batch; FULL: all 32 points live (unconditional stores: one basic block, so the
// compiler can count the stores that follow the loads of the next batch's jets instead of waiting for all of them)
template <bool FULL>
__device__ __forceinline__ void few_inputs_products(const float* stg, const int* stg_slot, const float* sT, const float* sB,
                                                    float* __restrict__ ytile, int nlive, int I, int OT, int o0, int lane,
                                                    float (&amax)[4], bool want_amax, const f32x4& touch) {
  const int pl = lane >> 3, oc = lane & 7;
#pragma unroll
  for (int st = 0; st < kFewBatch / 8; ++st) {
    const int pt = st * 8 + pl;
    f32x2 tl[8], th[8];                      // jets (0,1) and (2,3) of outputs oc + 8j
#pragma unroll
    for (int j = 0; j < 8; ++j) { tl[j] = (f32x2){0.f, 0.f}; th[j] = (f32x2){0.f, 0.f}; }
#pragma unroll
    for (int f = 0; f < kFewInputsMax; ++f) {
      if (f < I) {
        const float* e = stg + (pt * I + f) * kFewPair;
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(e), e1 = *reinterpret_cast<const f32x4*>(e + 4),
                    e2 = *reinterpret_cast<const f32x4*>(e + 8), e3 = *reinterpret_cast<const f32x4*>(e + 12),
                    e4 = *reinterpret_cast<const f32x4*>(e + 16);
        const int slot = stg_slot[pt * I + f];
        const float* tb = sT + (f * 64 + oc) * kFewRow + slot * 4;
        const f32x4 bw0 = *reinterpret_cast<const f32x4*>(sB + f * 64 + oc * 8);
        const f32x4 bw1 = *reinterpret_cast<const f32x4*>(sB + f * 64 + oc * 8 + 4);
        // e0 = (el0, el1), e1 = (el2, el3), e2 = (el4, eh0), e3 = (eh1, eh2), e4 = (eh3, eh4)
        const f32x2 el0 = lo2(e0), el1 = hi2(e0), el2 = lo2(e1), el3 = hi2(e1), el4 = lo2(e2);
        const f32x2 eh0 = hi2(e2), eh1 = lo2(e3), eh2 = hi2(e3), eh3 = lo2(e4), eh4 = hi2(e4);
        // (weights 4 outputs at a time: with four waves per SIMD the LDS latency hides behind the other waves, and
        // all 8 in flight at once do not fit the 128 registers a wave has)
        f32x4 w[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const f32x4*>(tb + 8 * j * kFewRow);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (j == 4) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 4; jj < 8; ++jj) w[jj] = *reinterpret_cast<const f32x4*>(tb + 8 * jj * kFewRow);
          }
          const f32x2 w01 = lo2(w[j]), w23 = hi2(w[j]);
          const f32x4 bw = j < 4 ? bw0 : bw1;
          const f32x2 wb = (j & 2) ? hi2(bw) : lo2(bw);
          pkfma_x(tl[j], el0, w01); pkfma_x(th[j], eh0, w01);
          pkfma_y(tl[j], el1, w01); pkfma_y(th[j], eh1, w01);
          pkfma_x(tl[j], el2, w23); pkfma_x(th[j], eh2, w23);
          pkfma_y(tl[j], el3, w23); pkfma_y(th[j], eh3, w23);
          if (j & 1) { pkfma_y(tl[j], el4, wb); pkfma_y(th[j], eh4, wb); }
          else       { pkfma_x(tl[j], el4, wb); pkfma_x(th[j], eh4, wb); }
        }
      }
    }
    if (want_amax && (FULL || pt < nlive)) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        absmax3_into(amax[0], tl[j][0], tl[j + 1][0]); absmax3_into(amax[1], tl[j][1], tl[j + 1][1]);
        absmax3_into(amax[2], th[j][0], th[j + 1][0]); absmax3_into(amax[3], th[j][1], th[j + 1][1]);
      }
    }
    // Loads and stores retire in one order: a wait for the next batch's jets placed behind this batch's stores is a
    // wait for the stores to reach memory.  Touching the jets before the second half of the stores puts the wait
    // there, most of a batch of arithmetic after the loads went out.
    if (st == kFewBatch / 8 - 2) asm volatile("" :: "v"(touch));
    float* dst = ytile + (pt * OT + o0 + oc) * 4;
    if (FULL || pt < nlive) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        *reinterpret_cast<f32x4*>(dst + j * 32) = (f32x4){tl[j][0], tl[j][1], th[j][0], th[j][1]};
    }
  }
}

__global__ void __launch_bounds__(64 * kFewWaves, 1)
fwd_sparse_few_inputs_kernel(const float* __restrict__ a, const float* __restrict__ grid, const float* __restrict__ ginfo,
                             const float* __restrict__ win, const float* __restrict__ wbt, float* __restrict__ y, int64_t N,
                             int I, int G, int act, int OT, int skip_near_uniform, unsigned* __restrict__ amax_out,
                             SeedSrc seed) {
  if (skip_near_uniform && all_rows_near_uniform(ginfo, I)) return;   // the fp16 matrix kernel beside this one takes the layer
  const int o0 = (int)blockIdx.y * 64;
  const int nk = G + 7;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = reinterpret_cast<float*>(smem);            // [I][64][kFewRow]
  float* sB = sT + kFewInputsMax * 64 * kFewRow;         // [I][8 (output mod 8)][8 (output / 8)]
  float* sK = sB + kFewInputsMax * 64;                   // [I][kKnotPadS]
  float* sE = sK + kFewInputsMax * kKnotPadS;            // [16 waves][64 pairs][kFewPair], then [16 waves][64] slots
  int* sS = reinterpret_cast<int*>(sE + kFewWaves * 64 * kFewPair);

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  for (int t = tid; t < I * (nk + 1); t += 64 * kFewWaves) {   // knot rows + the feature's 1/h (0 = not equally spaced)
    const int f = t / (nk + 1), k = t - f * (nk + 1);
    sK[f * kKnotPadS + k] = (k < nk) ? grid[(size_t)f * nk + k] : (ginfo ? ginfo[f] : 0.0f);
  }
  for (int t = tid; t < I * 64 * 16; t += 64 * kFewWaves) {    // window tables, 16 bytes at a time
    const int i = t >> 10, r = (t >> 4) & 63, c = t & 15;
    *reinterpret_cast<f32x4*>(sT + (i * 64 + r) * kFewRow + c * 4) =
        *reinterpret_cast<const f32x4*>(win + ((size_t)(i * OT + o0 + r) * 64 + c * 4));
  }
  for (int t = tid; t < I * 64; t += 64 * kFewWaves) {
    const int i = t >> 6, o = t & 63;
    sB[i * 64 + (o & 7) * 8 + (o >> 3)] = wbt[(size_t)i * OT + o0 + o];
  }
  __syncthreads();

  float* stg = sE + wave * (64 * kFewPair);
  int* stg_slot = sS + wave * 64;
  // evaluation: lane = pair (point, feature) of the batch
  const int ept = (I == 2) ? (lane >> 1) : lane, ef = (I == 2) ? (lane & 1) : 0;
  const bool eval_lane = ept < kFewBatch;
  // a block walks a contiguous range of batches, its waves side by side: what the chip writes at any moment is then
  // one compact window per block (16 x 32 KiB), not single batches spread over the whole tensor
  const int64_t nbatch_all = (N + kFewBatch - 1) / kFewBatch;
  const int64_t per_block = (nbatch_all + gridDim.x - 1) / gridDim.x;
  const int64_t bfirst = (int64_t)blockIdx.x * per_block;
  const int64_t nbatch = (bfirst + per_block < nbatch_all) ? bfirst + per_block : nbatch_all;
  const int64_t bstride = kFewWaves;
  int64_t b = bfirst + wave;
  float amax[4] = {0.f, 0.f, 0.f, 0.f};                  // per-jet maxima of the outputs this lane produced
  // jets of a batch: a wave-uniform row address (the batch's first point, or the last point of all for a batch past
  // the end) plus a 32-bit offset per lane
  f32x4 nxt;
  // (seed.x given: the jets are built from the points, value + unit derivative in the feature's own direction, and
  // left at seed.a_out for the backward pass: the first layer of a model)
#define PDEKAN_FEW_FETCH(B)                                                                          \
  {                                                                                                  \
    const int nl_ = tile_live((B) * kFewBatch, N, kFewBatch);                                        \
    const int64_t first_ = nl_ > 0 ? (B) * kFewBatch : N - 1;                                        \
    const int r_ = (nl_ > 0 && eval_lane) ? (ept < nl_ ? ept : nl_ - 1) : 0;                         \
    if (seed.x) {                                                                                    \
      const float v_ = seed.x[(first_ + r_) * seed.sn + ef * seed.sd];                               \
      nxt = (f32x4){v_, ef == 0 ? 1.0f : 0.0f, ef == 1 ? 1.0f : 0.0f, 0.0f};                         \
    } else {                                                                                         \
      nxt = *reinterpret_cast<const f32x4*>(a + first_ * I * 4 + (r_ * I + ef) * 4);                 \
    }                                                                                                \
  }
  PDEKAN_FEW_FETCH(b)
  for (; b < nbatch; b += bstride) {
    const int64_t P0 = b * kFewBatch;
    const int nlive = tile_live(P0, N, kFewBatch);
    const f32x4 cur = nxt;
    PDEKAN_FEW_FETCH(b + bstride)
    if (seed.x && eval_lane && ept < nlive && blockIdx.y == 0)
      *reinterpret_cast<f32x4*>(seed.a_out + ((P0 + ept) * I + ef) * 4) = cur;
    if (eval_lane) {
      const Live L = eval_live<false>(sK + ef * kKnotPadS, nk, act, cur[0]);
      const float s2 = cur[1] * cur[1];
      f32x2 el[5], eh[5];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        el[m] = (f32x2){L.b[m], L.d1[m] * cur[1]};
        eh[m] = (f32x2){L.d1[m] * cur[2], L.d2[m] * s2 + L.d1[m] * cur[3]};
      }
      el[4] = (f32x2){L.act.f0, L.act.f1 * cur[1]};
      eh[4] = (f32x2){L.act.f1 * cur[2], L.act.f2 * s2 + L.act.f1 * cur[3]};
      float* e = stg + lane * kFewPair;
      *reinterpret_cast<f32x4*>(e) = (f32x4){el[0][0], el[0][1], el[1][0], el[1][1]};
      *reinterpret_cast<f32x4*>(e + 4) = (f32x4){el[2][0], el[2][1], el[3][0], el[3][1]};
      *reinterpret_cast<f32x4*>(e + 8) = (f32x4){el[4][0], el[4][1], eh[0][0], eh[0][1]};
      *reinterpret_cast<f32x4*>(e + 12) = (f32x4){eh[1][0], eh[1][1], eh[2][0], eh[2][1]};
      *reinterpret_cast<f32x4*>(e + 16) = (f32x4){eh[3][0], eh[3][1], eh[4][0], eh[4][1]};
      stg_slot[lane] = L.slot;
    }
    float* ytile = y + P0 * OT * 4;
    if (nlive == kFewBatch) few_inputs_products<true>(stg, stg_slot, sT, sB, ytile, nlive, I, OT, o0, lane, amax, amax_out != nullptr, nxt);
    else few_inputs_products<false>(stg, stg_slot, sT, sB, ytile, nlive, I, OT, o0, lane, amax, amax_out != nullptr, nxt);
  }
#undef PDEKAN_FEW_FETCH
  if (amax_out) publish_jet_absmax(amax[0], amax[1], amax[2], amax[3], amax_out);
}

// ==========================================================================
// input-jet gradients
// ==========================================================================
template <int O, int RS = 64, bool MG = (RS != 64)>
__global__ void __launch_bounds__(256, 2)
bwd_input_sparse_kernel(const float* __restrict__ a, const float* __restrict__ gy, const float* __restrict__ grid,
                        const float* __restrict__ ginfo, const float* __restrict__ win, const float* __restrict__ wbt, float* __restrict__ ga,
                        int64_t N, int I, int G, int act, int OTr, int skip_near_uniform, unsigned* __restrict__ cot_out) {
  if (skip_near_uniform && all_rows_near_uniform(ginfo, I)) return;   // the fp16 matrix kernel beside this one takes the layer
  constexpr int OG = O / 32;
  constexpr int PW = 64 / OG;
  constexpr int kMaxSlot = (RS == 64) ? kSlots - 1 : kRawRow - 4;
  // MG: the layer's OT outputs are walked O at a time, each pass adding its share to the input gradients
  const int OT = MG ? OTr : O;
  const int npass = MG ? OTr / O : 1;
  const int nk = G + 7;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = reinterpret_cast<float*>(smem);
  float* sB = sT + 2 * O * RS;
  float* sJ = sB + 2 * O;                                // [4 waves][2][PW][kFS][4] staged input jets
  float* sO = sJ + 4 * 2 * PW * kFS * 4;                 // [4 waves][PW][kFS][4] staged output gradients
  float* sK = sO + 4 * PW * kFS * 4;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int g = lane / PW, pl = lane % PW;
  const int64_t P0 = ((int64_t)blockIdx.x * 4 + wave) * PW;
  int64_t p = P0 + pl;
  const bool live_pt = p < N;
  if (!live_pt) p = N - 1;
  float* sJw = sJ + wave * 2 * PW * kFS * 4;
  float* sOw = sO + wave * PW * kFS * 4;
  // write the kFS gradients parked for features i0..i0+3: 4 lanes store the 64 contiguous bytes of a point
  const WaveRows<PW> wr = wave_rows<PW>(P0, N, I, lane);
  f32x4 gmax = {0.f, 0.f, 0.f, 0.f};       // per-jet maxima of what the last pass stores (for cot_out)
  auto flush_out = [&](int i0, bool add, bool last_pass) {
    float* dst = ga + (wr.base * I + i0) * 4;
#pragma unroll
    for (int k = 0; k < WaveRows<PW>::NV; ++k) {
      f32x4 v = *reinterpret_cast<const f32x4*>(sOw + (lane + 64 * k) * 4);
      if ((lane + 64 * k) / kFS < wr.nlive) {
        // a later pass adds to what this same lane stored at this address in the pass before
        if (MG && add) v += *reinterpret_cast<const f32x4*>(dst + wr.off[k]);
        *reinterpret_cast<f32x4*>(dst + wr.off[k]) = v;
        if (last_pass) {
#pragma unroll
          for (int j = 0; j < 4; ++j) gmax[j] = fmaxf(gmax[j], fabsf(v[j]));
        }
      }
    }
  };

  for (int t = tid; t < I * (nk + 1); t += 256) {        // knot rows + the feature's 1/h (0 = not equally spaced)
    const int f = t / (nk + 1), k = t - f * (nk + 1);
    sK[f * kKnotPadS + k] = (k < nk) ? grid[(size_t)f * nk + k] : (ginfo ? ginfo[f] : 0.0f);
  }
  for (int pass = 0; pass < npass; ++pass) {
    const int o0 = pass * O;
    if (pass > 0) __syncthreads();                       // every wave is done with the previous pass's tables
    // cotangents of this lane's 32 outputs stay in registers for the whole kernel.  They are 512
    // contiguous bytes per lane in gy; fetched as whole 128-byte lines (8 lanes per line) and handed to
    // their lanes through LDS, 8 outputs at a time (the staging buffers are still unused here).
    f32x2 gyl[32], gyh[32];
    {
      constexpr int LS = 36;                               // floats per parked line: 32 + 4 pad (bank spread)
      float* park = sT + wave * (64 * LS);
      static_assert(4 * 64 * LS <= 2 * O * RS + 2 * O + 4 * 2 * PW * kFS * 4 + 4 * PW * kFS * 4,
                    "parking area fits the staging buffers (it ends before the knot rows)");
      const int line_w = pl * OG + g;                      // the line this lane consumes: (point, output group)
  #pragma unroll
      for (int k = 0; k < 4; ++k) {
  #pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int line = (lane >> 3) + 8 * r, j = lane & 7;
          int64_t pp = P0 + line / OG;
          if (pp > N - 1) pp = N - 1;
          *reinterpret_cast<f32x4*>(park + line * LS + j * 4) =
              *reinterpret_cast<const f32x4*>(gy + (pp * OT + o0 + (line % OG) * 32 + 8 * k + j) * 4);
        }
  #pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(park + line_w * LS + j * 4);
          gyl[8 * k + j] = lo2(v); gyh[8 * k + j] = hi2(v);
        }
      }
    }
    __syncthreads();                                       // parking area released before the tables land in it
    table_stage<O, RS>(sT, sB, win, wbt, 0, wave, lane, OT, o0);
    jets_stage<PW>(sJw, a, wr, I, 0);

    // as in the forward kernel, the basis of feature i+1 is evaluated before the barrier that ends iteration i
    barrier_after_dma();
    f32x4 cur = *reinterpret_cast<const f32x4*>(sJw + (pl * kFS) * 4);
    Live L = eval_live<true, kMaxSlot>(sK, nk, act, cur[0]);

    for (int i = 0; i < I; ++i) {
      const int buf = i & 1;
      if (i + 1 < I)
        table_stage<O, RS>(sT + (buf ^ 1) * O * RS, sB + (buf ^ 1) * O, win, wbt, i + 1, wave, lane, OT, o0);
      if ((i % kFS) == 0) {
        if (i > 0) flush_out(i - kFS, pass > 0, cot_out && pass == npass - 1);   // gradients of the previous 4 features, parked before the barrier
        if (i + kFS < I) jets_stage<PW>(sJw + (((i / kFS) + 1) & 1) * PW * kFS * 4, a, wr, I, i + kFS);
      }
      const float* tb = sT + buf * O * RS + (g * 32) * RS + L.slot * (RS == 64 ? 4 : 1);
      const float* bb = sB + buf * O + g * 32;
      f32x2 Gl[5], Gh[5];                    // per live column: sums over outputs of w * cotangent, jets (0,1) | (2,3)
  #pragma unroll
      for (int m = 0; m < 5; ++m) { Gl[m] = (f32x2){0.f, 0.f}; Gh[m] = (f32x2){0.f, 0.f}; }
      f32x4 wq[2][4], bq[2];
  #pragma unroll
      for (int k = 0; k < 4; ++k) wq[0][k] = live_weights<RS>(tb + k * RS);
      bq[0] = *reinterpret_cast<const f32x4*>(bb);
  #pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int cw = q & 1, nw = cw ^ 1;
        if (q + 1 < 8) {
  #pragma unroll
          for (int k = 0; k < 4; ++k) wq[nw][k] = live_weights<RS>(tb + ((q + 1) * 4 + k) * RS);
          bq[nw] = *reinterpret_cast<const f32x4*>(bb + (q + 1) * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x2 w01 = lo2(wq[cw][k]), w23 = hi2(wq[cw][k]);
          const f32x2 wb = (k < 2) ? lo2(bq[cw]) : hi2(bq[cw]);
          const f32x2 vl = gyl[q * 4 + k], vh = gyh[q * 4 + k];
          pkfma_x(Gl[0], vl, w01); pkfma_x(Gh[0], vh, w01);
          pkfma_y(Gl[1], vl, w01); pkfma_y(Gh[1], vh, w01);
          pkfma_x(Gl[2], vl, w23); pkfma_x(Gh[2], vh, w23);
          pkfma_y(Gl[3], vl, w23); pkfma_y(Gh[3], vh, w23);
          if (k & 1) { pkfma_y(Gl[4], vl, wb); pkfma_y(Gh[4], vh, wb); }
          else       { pkfma_x(Gl[4], vl, wb); pkfma_x(Gh[4], vh, wb); }
        }
      }
      // chain rule on this lane's partial sums (linear in G, so it commutes with the sum over groups)
      const float s2 = cur[1] * cur[1];
      f32x4 out = {0.f, 0.f, 0.f, 0.f};
  #pragma unroll
      for (int m = 0; m < 5; ++m) {
        const float e1 = m < 4 ? L.d1[m & 3] : L.act.f1;
        const float e2 = m < 4 ? L.d2[m & 3] : L.act.f2;
        const float e3 = m < 4 ? L.d3[m & 3] : L.act.f3;
        const f32x4 q = {Gl[m][0], Gl[m][1], Gh[m][0], Gh[m][1]};
        out[0] += q[0] * e1 + (q[1] * cur[1] + q[2] * cur[2]) * e2 + q[3] * (s2 * e3 + cur[3] * e2);
        out[1] += q[1] * e1 + 2.0f * cur[1] * q[3] * e2;
        out[2] += q[2] * e1;
        out[3] += q[3] * e1;
      }
      // add the output groups of one point (lanes pl, pl+PW, ...)
      if (PW == 16) {
  #pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = sum_xor16(out[j]);
      }
  #pragma unroll
      for (int j = 0; j < 4; ++j) out[j] = sum_xor32(out[j]);
      if (g == 0) *reinterpret_cast<f32x4*>(sOw + (pl * kFS + (i % kFS)) * 4) = out;
      if (i + 1 < I) {
        const int n = i + 1;
        cur = *reinterpret_cast<const f32x4*>(sJw + (((n / kFS) & 1) * PW * kFS + pl * kFS + (n % kFS)) * 4);
        L = eval_live<true, kMaxSlot>(sK + n * kKnotPadS, nk, act, cur[0]);
        barrier_after_dma();
      }
    }
    __syncthreads();
    flush_out(I - kFS, pass > 0, cot_out && pass == npass - 1);
  }
  if (cot_out) publish_jet_absmax(gmax[0], gmax[1], gmax[2], gmax[3], cot_out);
}

// ==========================================================================
// host side
// ==========================================================================
// Window tables (one aligned 16-byte LDS read per output) need G + 6 <= 16 slots and 64 or 128 outputs; every other
// layer up to G = 20 and 256 outputs takes the raw-row tables (64 outputs per block, two ds_read2_b32 per output).
static bool window_layout(const LayerView& L) {
  const char* raw = getenv("PDEKAN_FORCE_RAW_ROWS");     // A/B switch: raw rows also where a window table would fit
  if (raw && raw[0] == '1' && L.I >= 8 && L.I % kFS == 0) return false;
  return L.G + 6 <= kSlots && (L.O == 64 || L.O == 128);
}
static bool raw_layout(const LayerView& L) {
  return !window_layout(L) && L.G + 10 <= kRawRow && L.O % 64 == 0 && L.O <= 256 && L.I >= 8 && L.I % kFS == 0;
}

bool sparse_eligible(int D, int D2, const LayerView& L) {
  if (D != 2 || D2 != 1) return false;
  const char* off = getenv("PDEKAN_DISABLE_SPARSE");
  if (off && off[0] == '1') return false;
  if (L.I < 2 || L.I > 256) return false;
  if (raw_layout(L)) return true;
  if (!window_layout(L)) return false;
  // the forward kernel stages 4 features of a point at a time, or 2 (64 outputs only)
  return (L.I % 4 == 0) || (L.O == 64 && L.I % 2 == 0);
}

// the input-gradient kernel stages 4 features at a time
bool sparse_input_grad_eligible(int D, int D2, const LayerView& L) {
  return sparse_eligible(D, D2, L) && L.I >= 8 && L.I % kFS == 0;
}

size_t sparse_table_floats(const LayerView& L) {
  return (size_t)L.I * L.O * ((window_layout(L) ? kSlots * 4 : kRawRow) + 1);
}

void launch_pack_window(const LayerView& L, const float* wp, float* win, hipStream_t st) {
  const bool window = window_layout(L);
  const int per = window ? kSlots : kRawRow;
  const int total = L.I * L.O * per;
  int blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  float* wbt = win + (size_t)L.I * L.O * (window ? kSlots * 4 : kRawRow);
  if (window) pack_window_kernel<<<blocks, 256, 0, st>>>(wp, win, wbt, L.I, L.O, L.C());
  else pack_raw_rows_kernel<<<blocks, 256, 0, st>>>(wp, win, wbt, L.I, L.O, L.C());
}

// dynamic LDS of a block: tables, base weights, staged jets (+ parked gradients), knot rows; never less than the
// area the wide tensors are parked in on their way to / from registers (4 waves x 64 lines x 36 floats)
template <int O, int RS = 64>
static size_t sparse_lds(const LayerView& L, bool with_out_stage, int fs = kFS) {
  constexpr int PW = 64 / (O / 32);
  const size_t floats = (size_t)(2 * O * RS + 2 * O + 4 * 2 * PW * fs * 4 + (with_out_stage ? 4 * PW * fs * 4 : 0) +
                                 L.I * kKnotPadS);
  const size_t park = 4 * 64 * 36;
  return (floats > park ? floats : park) * 4;
}

template <typename Kern>
static void big_lds(Kern k, size_t bytes) {
  if (bytes > 48 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(k), bytes);
}

template <int O, int FS, int RS = 64, bool MG = (RS != 64)>
static void fwd_sparse_launch(const LayerView& L, const float* a, const float* win, float* y, int64_t N, hipStream_t st,
                              int skip, unsigned* amax_out) {
  const float* wbt = win + (size_t)L.I * L.O * RS;
  constexpr int pts = 4 * (64 / (O / 32));               // points per block
  auto k = fwd_sparse_kernel<O, FS, RS, MG>;
  const size_t lds = sparse_lds<O, RS>(L, false, FS);
  big_lds(k, lds);
  const dim3 g((unsigned)((N + pts - 1) / pts), MG ? L.O / O : 1);
  k<<<g, 256, lds, st>>>(a, L.grid, L.ginfo, win, wbt, y, N, L.I, L.G, L.act, L.O, skip, amax_out);
}

// one or two inputs, window tables: one resident block per compute unit (fwd_sparse_few_inputs_kernel)
static bool few_inputs_layout(const LayerView& L) {
  const char* off = getenv("PDEKAN_DISABLE_FEW_INPUTS");  // A/B switch
  if (off && off[0] == '1') return false;
  return window_layout(L) && L.I <= kFewInputsMax;
}

static void fwd_sparse_few_inputs_launch(const LayerView& L, const float* a, const float* win, float* y, int64_t N,
                                         hipStream_t st, int skip, unsigned* amax_out, SeedSrc seed) {
  const float* wbt = win + (size_t)L.I * L.O * 64;
  auto k = fwd_sparse_few_inputs_kernel;
  const size_t lds = ((size_t)kFewInputsMax * 64 * kFewRow + kFewInputsMax * 64 + kFewInputsMax * kKnotPadS +
                      (size_t)kFewWaves * 64 * kFewPair + kFewWaves * 64) * 4;
  big_lds(k, lds);
  const int64_t batches = (N + kFewBatch - 1) / kFewBatch;
  int64_t blocks = (batches + kFewWaves - 1) / kFewWaves;
  const int64_t resident = 256 / (L.O / 64);             // one block per compute unit
  if (blocks > resident) blocks = resident;
  if (blocks < 1) blocks = 1;
  const dim3 g((unsigned)blocks, L.O / 64);
  k<<<g, 64 * kFewWaves, lds, st>>>(a, L.grid, L.ginfo, win, wbt, y, N, L.I, L.G, L.act, L.O, skip, amax_out, seed);
}

bool fwd_sparse_takes_seed(const LayerView& L) { return few_inputs_layout(L); }

// seed (x != nullptr): only for layers fwd_sparse_takes_seed() accepts
void launch_fwd_sparse(const LayerView& L, const float* a, const float* win, float* y, int64_t N, hipStream_t st,
                       bool skip_near_uniform, unsigned* amax_out, SeedSrc seed) {
  const int skip = skip_near_uniform ? 1 : 0;
  if (few_inputs_layout(L)) {
    fwd_sparse_few_inputs_launch(L, a, win, y, N, st, skip, amax_out, seed);
  } else if (!window_layout(L)) {
    fwd_sparse_launch<64, 4, kRawRow>(L, a, win, y, N, st, skip, amax_out);
  } else if (L.O == 64) {
    if (L.I % 4 == 0) fwd_sparse_launch<64, 4>(L, a, win, y, N, st, skip, amax_out);
    else fwd_sparse_launch<64, 2>(L, a, win, y, N, st, skip, amax_out);
  } else {                                               // 128 outputs: two groups of 64 (faster than one lane group of 4 x 32)
    fwd_sparse_launch<64, 4, 64, true>(L, a, win, y, N, st, skip, amax_out);
  }
}

template <int O, int RS = 64, bool MG = (RS != 64)>
static void bwd_input_sparse_launch(const LayerView& L, const float* a, const float* gy, const float* win, float* ga,
                                    int64_t N, hipStream_t st, int skip, unsigned* cot_out) {
  const float* wbt = win + (size_t)L.I * L.O * RS;
  constexpr int pts = 4 * (64 / (O / 32));
  auto k = bwd_input_sparse_kernel<O, RS, MG>;
  const size_t lds = sparse_lds<O, RS>(L, true);
  big_lds(k, lds);
  k<<<(unsigned)((N + pts - 1) / pts), 256, lds, st>>>(a, gy, L.grid, L.ginfo, win, wbt, ga, N, L.I, L.G, L.act, L.O, skip, cot_out);
}

void launch_bwd_input_sparse(const LayerView& L, const float* a, const float* gy, const float* win, float* ga,
                             int64_t N, hipStream_t st, bool skip_near_uniform, unsigned* cot_out) {
  const int skip = skip_near_uniform ? 1 : 0;
  if (!window_layout(L)) bwd_input_sparse_launch<64, kRawRow>(L, a, gy, win, ga, N, st, skip, cot_out);
  else if (L.O == 64) bwd_input_sparse_launch<64>(L, a, gy, win, ga, N, st, skip, cot_out);
  else bwd_input_sparse_launch<64, 64, true>(L, a, gy, win, ga, N, st, skip, cot_out);
}

}  // namespace pdekan
