// Host build of pde_amd/csrc/kan_math.h for CPU-side unit tests (test
// infrastructure; never loaded by the product).  The loops below restate what
// one thread of the streaming kernels does, calling the very same inline math
// the device code calls, so the formulas can be checked against the oracle and
// the golden fixtures without a GPU.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../pde_amd/csrc/kan_math.h"

using namespace pdekan;

template <int D, int D2>
static void fwd_t(const float* a, const float* grid, const float* wp, float* y, int64_t N, int I, int O, int G,
                  int act) {
  constexpr int J = 1 + D + D2;
  const int nk = G + 7, C = G + 3, C1 = C + 1;
  for (int64_t p = 0; p < N; ++p)
    for (int o = 0; o < O; ++o) {
      float acc[J];
      for (int j = 0; j < J; ++j) acc[j] = 0.f;
      for (int i = 0; i < I; ++i) {
        const float* jet = a + (p * I + i) * J;
        Basis bs;
        eval_basis<false>(grid + i * nk, nk, jet[0], bs);
        const Act av = eval_act(act, jet[0]);
        const float* w = wp + ((size_t)o * I + i) * C1;
        float p0 = w[C] * av.f0, p1 = w[C] * av.f1, p2 = w[C] * av.f2;
        for (int m = 0; m < 4; ++m) {
          const int c = bs.c0 + m;
          if (c >= 0 && c < C) { p0 += w[c] * bs.b[m]; p1 += w[c] * bs.d1[m]; p2 += w[c] * bs.d2[m]; }
        }
        jet_accumulate<D, D2>(acc, jet, p0, p1, p2);
      }
      for (int j = 0; j < J; ++j) y[(p * O + o) * J + j] = acc[j];
    }
}

template <int D, int D2>
static void bwd_t(const float* a, const float* gy, const float* grid, const float* wp, float* ga, float* gw,
                  int64_t N, int I, int O, int G, int act) {
  constexpr int J = 1 + D + D2;
  const int nk = G + 7, C = G + 3, C1 = C + 1;
  std::vector<double> gwd((size_t)O * I * C1, 0.0);
  for (int64_t p = 0; p < N; ++p)
    for (int i = 0; i < I; ++i) {
      const float* jet = a + (p * I + i) * J;
      Basis bs;
      eval_basis<true>(grid + i * nk, nk, jet[0], bs);
      const Act av = eval_act(act, jet[0]);
      float Gm[5][J];
      for (int m = 0; m < 5; ++m) for (int j = 0; j < J; ++j) Gm[m][j] = 0.f;
      for (int o = 0; o < O; ++o) {
        const float* w = wp + ((size_t)o * I + i) * C1;
        const float* g = gy + (p * O + o) * J;
        float A0, A1, A2;
        jet_pullback_coeffs<D, D2>(g, jet, A0, A1, A2);
        gwd[((size_t)o * I + i) * C1 + C] += A0 * av.f0 + A1 * av.f1 + A2 * av.f2;
        for (int m = 0; m < 4; ++m) {
          const int c = bs.c0 + m;
          if (c < 0 || c >= C) continue;
          gwd[((size_t)o * I + i) * C1 + c] += A0 * bs.b[m] + A1 * bs.d1[m] + A2 * bs.d2[m];
          for (int j = 0; j < J; ++j) Gm[m][j] += w[c] * g[j];
        }
        for (int j = 0; j < J; ++j) Gm[4][j] += w[C] * g[j];
      }
      float out[J];
      for (int j = 0; j < J; ++j) out[j] = 0.f;
      for (int m = 0; m < 5; ++m) {
        const float e1 = m < 4 ? bs.d1[m & 3] : av.f1, e2 = m < 4 ? bs.d2[m & 3] : av.f2,
                    e3 = m < 4 ? bs.d3[m & 3] : av.f3;
        out[0] += Gm[m][0] * e1;
        for (int d = 0; d < D; ++d) { out[0] += Gm[m][1 + d] * jet[1 + d] * e2; out[1 + d] += Gm[m][1 + d] * e1; }
        for (int d = 0; d < D2; ++d) {
          const float gdd = Gm[m][1 + D + d];
          out[0] += gdd * (jet[1 + d] * jet[1 + d] * e3 + jet[1 + D + d] * e2);
          out[1 + d] += 2.f * jet[1 + d] * gdd * e2;
          out[1 + D + d] += gdd * e1;
        }
      }
      if (ga) for (int j = 0; j < J; ++j) ga[(p * I + i) * J + j] = out[j];
    }
  if (gw) for (size_t k = 0; k < gwd.size(); ++k) gw[k] = (float)gwd[k];
}

extern "C" {

void hm_eval_basis(const float* knots, int nk, float x, float* out16, int* c0, int* inside) {
  Basis bs;
  eval_basis<true>(knots, nk, x, bs);
  for (int m = 0; m < 4; ++m) { out16[m] = bs.b[m]; out16[4 + m] = bs.d1[m]; out16[8 + m] = bs.d2[m]; out16[12 + m] = bs.d3[m]; }
  *c0 = bs.c0; *inside = bs.inside;
}

// knots followed by the 1/h slot, as the kernels stage them; uniform != 0 takes the equal-spacing path
void hm_eval_basis_row(const float* knots, int nk, int uniform, float x, float* out16, int* c0, int* inside) {
  std::vector<float> row(knots, knots + nk);
  const float h = (knots[nk - 1] - knots[0]) / (float)(nk - 1);
  row.push_back(uniform == 1 ? 1.0f / h : uniform == 2 ? equal_spacing_inv_h(knots, nk) : 0.0f);
  Basis bs;
  eval_basis_row<true>(row.data(), nk, x, bs);
  for (int m = 0; m < 4; ++m) { out16[m] = bs.b[m]; out16[4 + m] = bs.d1[m]; out16[8 + m] = bs.d2[m]; out16[12 + m] = bs.d3[m]; }
  *c0 = bs.c0; *inside = bs.inside;
}

// the loop-free form of the equal-spacing path (what the split-bf16 weight-gradient kernel calls)
void hm_eval_basis_uniform_straight(const float* knots, int nk, float x, float* out16, int* c0, int* inside) {
  const float h = (knots[nk - 1] - knots[0]) / (float)(nk - 1);
  Basis bs;
  eval_basis_uniform_straight<true>(knots, nk, 1.0f / h, x, bs);
  for (int m = 0; m < 4; ++m) { out16[m] = bs.b[m]; out16[4 + m] = bs.d1[m]; out16[8 + m] = bs.d2[m]; out16[12 + m] = bs.d3[m]; }
  *c0 = bs.c0; *inside = bs.inside;
}

// the same with one round of knot reads (fp16 matrix input-gradient kernel)
void hm_eval_basis_uniform_oneshot(const float* knots, int nk, float x, float* out16, int* c0, int* inside) {
  const float h = (knots[nk - 1] - knots[0]) / (float)(nk - 1);
  Basis bs;
  eval_basis_uniform_oneshot<true>(knots, nk, knots[0], knots[nk - 1], 1.0f / h, x, bs);
  for (int m = 0; m < 4; ++m) { out16[m] = bs.b[m]; out16[4 + m] = bs.d1[m]; out16[8 + m] = bs.d2[m]; out16[12 + m] = bs.d3[m]; }
  *c0 = bs.c0; *inside = bs.inside;
}

// interval from the quotient alone (weight-gradient kernels: B, B', B'' only)
void hm_eval_basis_uniform_by_quotient(const float* knots, int nk, float x, float* out16, int* c0, int* inside) {
  const float h = (knots[nk - 1] - knots[0]) / (float)(nk - 1);
  Basis bs;
  eval_basis_uniform_by_quotient(knots, nk, knots[0], knots[nk - 1], 1.0f / h, x, bs);
  for (int m = 0; m < 4; ++m) { out16[m] = bs.b[m]; out16[4 + m] = bs.d1[m]; out16[8 + m] = bs.d2[m]; out16[12 + m] = bs.d3[m]; }
  *c0 = bs.c0; *inside = bs.inside;
}

float hm_equal_spacing_inv_h(const float* knots, int nk) { return equal_spacing_inv_h(knots, nk); }

float hm_knot_spacing_info(const float* knots, int nk) { return knot_spacing_info(knots, nk); }

// the general-knot evaluation of the matrix-core kernels (with the third derivative), on a row padded the way they stage it
void hm_eval_basis_near_uniform(const float* knots, int nk, float x, float* out16, int* c0, int* inside) {
  std::vector<float> row;
  for (int k = -3; k < nk + 4; ++k) row.push_back(knots[k < 0 ? 0 : (k > nk - 1 ? nk - 1 : k)]);
  const float h = (knots[nk - 1] - knots[0]) / (float)(nk - 1);
  Basis bs;
  eval_basis_near_uniform<true>(row.data() + 3, nk, knots[0], knots[nk - 1], 1.0f / h, x, bs);
  for (int m = 0; m < 4; ++m) { out16[m] = bs.b[m]; out16[4 + m] = bs.d1[m]; out16[8 + m] = bs.d2[m]; out16[12 + m] = bs.d3[m]; }
  *c0 = bs.c0; *inside = bs.inside;
}

void hm_eval_act(int kind, float x, float* out4) {
  const Act a = eval_act(kind, x);
  out4[0] = a.f0; out4[1] = a.f1; out4[2] = a.f2; out4[3] = a.f3;
}

int hm_layer_fwd(int D, int D2, const float* a, const float* grid, const float* wp, float* y, int64_t N, int I,
                 int O, int G, int act) {
  if (D == 0 && D2 == 0) fwd_t<0, 0>(a, grid, wp, y, N, I, O, G, act);
  else if (D == 2 && D2 == 1) fwd_t<2, 1>(a, grid, wp, y, N, I, O, G, act);
  else if (D == 6 && D2 == 5) fwd_t<6, 5>(a, grid, wp, y, N, I, O, G, act);
  else return -1;
  return 0;
}

int hm_layer_bwd(int D, int D2, const float* a, const float* gy, const float* grid, const float* wp, float* ga,
                 float* gw, int64_t N, int I, int O, int G, int act) {
  if (D == 0 && D2 == 0) bwd_t<0, 0>(a, gy, grid, wp, ga, gw, N, I, O, G, act);
  else if (D == 2 && D2 == 1) bwd_t<2, 1>(a, gy, grid, wp, ga, gw, N, I, O, G, act);
  else if (D == 6 && D2 == 5) bwd_t<6, 5>(a, gy, grid, wp, ga, gw, N, I, O, G, act);
  else return -1;
  return 0;
}

}  // extern "C"
