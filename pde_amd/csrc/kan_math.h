// Per-feature math of the KAN layer: local cubic B-spline evaluation with
// derivatives, base-activation derivatives, and the jet algebra that carries
// (value, first derivatives, diagonal second derivatives) through a layer.
//
// The reference evaluates all G+k bases of a feature densely by Cox-de Boor
// (efficient_kan/kan.py:78-111) and gets derivatives from autograd.  Only the
// k+1 = 4 bases whose support contains x are non-zero, so here one interval
// search is followed by the local de Boor triangle on the 6 knots around x and
// by the closed-form derivative recurrences
//     B'_{j,p} = p * ( B_{j,p-1}/(g_{j+p}-g_j) - B_{j+1,p-1}/(g_{j+p+1}-g_{j+1}) ).
// Interval convention is the reference's: half-open [g_s, g_{s+1}); a point
// outside [g_0, g_last) has every basis (and derivative) equal to zero.
//
// Everything is __host__ __device__ so the same code is unit-tested on the CPU.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define KAN_HD __host__ __device__ __forceinline__
#else
#define KAN_HD inline
#endif

namespace pdekan {

constexpr int kOrder = 3;            // cubic splines only (every reference caller uses spline_order=3)
constexpr int kActIdentity = 0;
constexpr int kActSiLU = 1;

KAN_HD float kan_rcp(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(v);
#else
  return 1.0f / v;
#endif
}

KAN_HD float kan_exp(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __expf(v);
#else
  return expf(v);
#endif
}

// Values and derivatives of the 4 non-zero cubic bases at x.
//   c0 : index of the first of them (may be negative or run past C-1 near the
//        ends of the extended grid; callers mask c0+m to [0, C-1]).
//   inside = 0 : x lies outside the grid (or is NaN); all arrays are zero.
struct Basis {
  float b[4];    // B_{c0+m,3}(x)
  float d1[4];   // first derivative
  float d2[4];   // second derivative
  float d3[4];   // third derivative (piecewise constant)
  int c0;
  int inside;
};

// Find s with g[s] <= x < g[s+1].  nk = number of knots.  A uniform guess is
// corrected by walking, so non-uniform knots (after update_grid) are handled.
template <typename KnotPtr>
KAN_HD int find_interval(const KnotPtr g, int nk, float x) {
  const float g0 = g[0], g1 = g[nk - 1];
  float u = (x - g0) * kan_rcp(g1 - g0) * (float)(nk - 1);
  int s = (u > 0.0f) ? (int)fminf(u, (float)(nk - 2)) : 0;   // NaN -> 0; clamped before the conversion
  while (s > 0 && x < g[s]) --s;
  while (s < nk - 2 && x >= g[s + 1]) ++s;
  return s;
}

template <bool kNeedD3, typename KnotPtr>
KAN_HD void eval_basis(const KnotPtr g, int nk, float x, Basis& out) {
  const int s = find_interval(g, nk, x);
  out.c0 = s - kOrder;
  out.inside = (x >= g[0]) && (x < g[nk - 1]);
  // knots g[s-2] .. g[s+3], indices clamped: a clamped knot only ever feeds a
  // basis whose index is outside [0, C-1], which callers drop.
  const int last = nk - 1;
  const float km2 = g[s - 2 < 0 ? 0 : s - 2];
  const float km1 = g[s - 1 < 0 ? 0 : s - 1];
  const float k0 = g[s];
  const float k1 = g[s + 1];
  const float k2 = g[s + 2 > last ? last : s + 2];
  const float k3 = g[s + 3 > last ? last : s + 3];

  const float r1 = kan_rcp(k1 - k0);
  const float r2a = kan_rcp(k1 - km1), r2b = kan_rcp(k2 - k0);
  const float r3a = kan_rcp(k1 - km2), r3b = kan_rcp(k2 - km1), r3c = kan_rcp(k3 - k0);

  // order 1 (two non-zero): indices s-1, s
  const float b1a = (k1 - x) * r1;
  const float b1b = (x - k0) * r1;
  // order 2 (three non-zero): indices s-2, s-1, s
  const float u_a = b1a * r2a, u_b = b1b * r2b;
  const float b2a = (k1 - x) * u_a;
  const float b2b = (x - km1) * u_a + (k2 - x) * u_b;
  const float b2c = (x - k0) * u_b;
  // order 3 (four non-zero): indices s-3 .. s
  const float q_a = b2a * r3a, q_b = b2b * r3b, q_c = b2c * r3c;
  float b0 = (k1 - x) * q_a;
  float b1 = (x - km2) * q_a + (k2 - x) * q_b;
  float b2 = (x - km1) * q_b + (k3 - x) * q_c;
  float b3 = (x - k0) * q_c;
  // first derivative: 3 * (q[j] - q[j+1])
  float e0 = -3.0f * q_a, e1 = 3.0f * (q_a - q_b), e2 = 3.0f * (q_b - q_c), e3 = 3.0f * q_c;
  // second derivative: order-2 first derivatives are 2*(u[j]-u[j+1])
  const float w_a = -2.0f * u_a * r3a, w_b = 2.0f * (u_a - u_b) * r3b, w_c = 2.0f * u_b * r3c;
  float f0 = -3.0f * w_a, f1 = 3.0f * (w_a - w_b), f2 = 3.0f * (w_b - w_c), f3 = 3.0f * w_c;
  const float m = out.inside ? 1.0f : 0.0f;   // outside the grid everything is zero
  out.b[0] = b0 * m; out.b[1] = b1 * m; out.b[2] = b2 * m; out.b[3] = b3 * m;
  out.d1[0] = e0 * m; out.d1[1] = e1 * m; out.d1[2] = e2 * m; out.d1[3] = e3 * m;
  out.d2[0] = f0 * m; out.d2[1] = f1 * m; out.d2[2] = f2 * m; out.d2[3] = f3 * m;
  if (kNeedD3) {
    const float z_a = 2.0f * r1 * r2a * r3a, z_b = -2.0f * r1 * (r2a + r2b) * r3b,
                z_c = 2.0f * r1 * r2b * r3c;
    out.d3[0] = -3.0f * z_a * m; out.d3[1] = 3.0f * (z_a - z_b) * m;
    out.d3[2] = 3.0f * (z_b - z_c) * m; out.d3[3] = 3.0f * z_c * m;
  } else {
    out.d3[0] = out.d3[1] = out.d3[2] = out.d3[3] = 0.0f;
  }
  if (!out.inside) {   // NaN/inf products above must not leak
    for (int i = 0; i < 4; ++i) { out.b[i] = out.d1[i] = out.d2[i] = out.d3[i] = 0.0f; }
  }
}

// Same outputs for a feature whose knots are equally spaced (the grid every
// KANLinear is built with, kan.py:27-35, until update_grid moves it): with
// t = (x - g_s) / h in [0,1) the four live cubic bases are fixed polynomials,
//   N0 = (1-t)^3/6, N1 = (3t^3-6t^2+4)/6, N2 = (-3t^3+3t^2+3t+1)/6, N3 = t^3/6,
// and each derivative order brings one factor 1/h.  No reciprocals, 2 knot reads.
template <bool kNeedD3>
KAN_HD void uniform_cubic_pieces(float t, bool inside, float inv_h, Basis& out) {
  const float m = inside ? 1.0f : 0.0f;
  const float t2 = t * t, t3 = t2 * t, q = 1.0f - t;
  const float sixth = 1.0f / 6.0f;
  out.b[0] = m * (q * q * q * sixth);
  out.b[1] = m * ((3.0f * t3 - 6.0f * t2 + 4.0f) * sixth);
  out.b[2] = m * ((-3.0f * t3 + 3.0f * t2 + 3.0f * t + 1.0f) * sixth);
  out.b[3] = m * (t3 * sixth);
  const float s1 = m * inv_h;
  out.d1[0] = s1 * (-0.5f * q * q);
  out.d1[1] = s1 * (1.5f * t2 - 2.0f * t);
  out.d1[2] = s1 * (-1.5f * t2 + t + 0.5f);
  out.d1[3] = s1 * (0.5f * t2);
  const float s2 = s1 * inv_h;
  out.d2[0] = s2 * q;
  out.d2[1] = s2 * (3.0f * t - 2.0f);
  out.d2[2] = s2 * (1.0f - 3.0f * t);
  out.d2[3] = s2 * t;
  if (kNeedD3) {
    const float s3 = s2 * inv_h;
    out.d3[0] = -s3; out.d3[1] = 3.0f * s3; out.d3[2] = -3.0f * s3; out.d3[3] = s3;
  } else {
    out.d3[0] = out.d3[1] = out.d3[2] = out.d3[3] = 0.0f;
  }
}

template <bool kNeedD3, typename KnotPtr>
KAN_HD void eval_basis_uniform(const KnotPtr g, int nk, float inv_h, float x, Basis& out) {
  const float g0 = g[0];
  float u = (x - g0) * inv_h;
  int s = (u > 0.0f) ? (int)fminf(u, (float)(nk - 2)) : 0;   // NaN -> 0; clamped before the conversion
  while (s > 0 && x < g[s]) --s;            // rounding at a knot: at most one step
  while (s < nk - 2 && x >= g[s + 1]) ++s;
  out.c0 = s - kOrder;
  out.inside = (x >= g0) && (x < g[nk - 1]);
  const float t = out.inside ? (x - g[s]) * inv_h : 0.0f;
  uniform_cubic_pieces<kNeedD3>(t, out.inside, inv_h, out);
}

// The same without a loop or branch (selects only), for code that must stay one
// basic block: the two rounding corrections above never take more than one step
// each when the knots are equally spaced.
template <bool kNeedD3, typename KnotPtr>
KAN_HD void eval_basis_uniform_straight(const KnotPtr g, int nk, float inv_h, float x, Basis& out) {
  const float g0 = g[0], glast = g[nk - 1];
  const float u = (x - g0) * inv_h;
  int s = (u > 0.0f) ? (int)fminf(u, (float)(nk - 2)) : 0;
  const float ga = g[s];                     // every knot read is unconditional: no short-circuit branches
  s -= (int)(s > 0) & (int)(x < ga);
  const float gb = g[s + 1];
  s += (int)(s < nk - 2) & (int)(x >= gb);
  const float gs = g[s];
  out.c0 = s - kOrder;
  out.inside = (int)(x >= g0) & (int)(x < glast);
  const float t = out.inside ? (x - gs) * inv_h : 0.0f;
  uniform_cubic_pieces<kNeedD3>(t, out.inside, inv_h, out);
}

// The same again with ONE round of knot reads: the three knots the two corrections can ask for are read at once
// (g[s-1], g[s], g[s+1] around the quotient guess s, indices clamped), then everything is selects in registers.
// Same interval and same outputs as eval_basis_uniform_straight.
template <bool kNeedD3, typename KnotPtr>
KAN_HD void eval_basis_uniform_oneshot(const KnotPtr g, int nk, float g0, float glast, float inv_h, float x, Basis& out) {
  const float u = (x - g0) * inv_h;
  int s = (u > 0.0f) ? (int)fminf(u, (float)(nk - 2)) : 0;
  const float ga = g[s > 0 ? s - 1 : 0], gb = g[s], gc = g[s + 1];
  const int down = (int)(s > 0) & (int)(x < gb);
  const int up = (int)(s < nk - 2) & (int)(x >= gc) & (down ^ 1);
  const float gs = down ? ga : (up ? gc : gb);
  s += up - down;
  out.c0 = s - kOrder;
  out.inside = (int)(x >= g0) & (int)(x < glast);
  const float t = out.inside ? (x - gs) * inv_h : 0.0f;
  uniform_cubic_pieces<kNeedD3>(t, out.inside, inv_h, out);
}

// For kernels that need B, B' and B'' only (weight gradients): the interval comes from the quotient alone,
// with g[0], g[nk-1] and 1/h passed in (registers), so the only knot read is g[s] and nothing depends on a
// comparison against a neighbouring knot.  When rounding puts x a hair outside [g_s, g_{s+1}) the piece of the
// neighbouring interval is evaluated a rounding error outside [0,1): B, B' and B'' are continuous across knots, so
// the values agree with the exact interval's to that rounding error.  The third derivative is NOT continuous
// there: out.d3 is left zero.
template <typename KnotPtr>
KAN_HD void eval_basis_uniform_by_quotient(const KnotPtr g, int nk, float g0, float glast, float inv_h, float x,
                                           Basis& out) {
  const float u = (x - g0) * inv_h;
  const int s = (u > 0.0f) ? (int)fminf(u, (float)(nk - 2)) : 0;
  const float gs = g[s];
  out.c0 = s - kOrder;
  out.inside = (int)(x >= g0) & (int)(x < glast);
  const float t = out.inside ? (x - gs) * inv_h : 0.0f;
  uniform_cubic_pieces<false>(t, out.inside, inv_h, out);
}

// B, B' and B'' on ANY knot row whose quotient guess floor((x - g_0)/h) is at most one interval off (see
// knot_spacing_info), again without a loop, a branch or a dependent chain of knot reads: the 8 knots around the
// guess are read at once, the interval is settled by two comparisons in registers and the general triangle of
// eval_basis runs on the 6 knots selected.  gp points at knot 0 of a row stored with its first knot repeated 3
// times in front and its last knot 4 times behind (the reads g[s-3] .. g[s+4] need no index clamp; a repeated
// knot only ever feeds a basis whose index is outside [0, C-1], which callers drop, and every denominator below
// spans [g_s, g_{s+1}], so none is zero).  The interval is settled by exact comparisons, so with kNeedD3 the third
// derivative (piecewise constant, jumping at the knots) is that of the reference's half-open interval; without it
// out.d3 is left zero.
template <bool kNeedD3 = false, typename KnotPtr>
KAN_HD void eval_basis_near_uniform(const KnotPtr gp, int nk, float g0, float glast, float inv_h, float x, Basis& out) {
  const float u = (x - g0) * inv_h;
  int s = (u > 0.0f) ? (int)fminf(u, (float)(nk - 2)) : 0;
  const float a0 = gp[s - 3], a1 = gp[s - 2], a2 = gp[s - 1], a3 = gp[s], a4 = gp[s + 1], a5 = gp[s + 2],
              a6 = gp[s + 3], a7 = gp[s + 4];
  const bool down = (s > 0) & (x < a3), up = (s < nk - 2) & (x >= a4);
  const float km2 = down ? a0 : (up ? a2 : a1);
  const float km1 = down ? a1 : (up ? a3 : a2);
  const float k0 = down ? a2 : (up ? a4 : a3);
  const float k1 = down ? a3 : (up ? a5 : a4);
  const float k2 = down ? a4 : (up ? a6 : a5);
  const float k3 = down ? a5 : (up ? a7 : a6);
  s += (int)up - (int)down;
  out.c0 = s - kOrder;
  out.inside = (int)(x >= g0) & (int)(x < glast);
  const float r1 = kan_rcp(k1 - k0);
  const float r2a = kan_rcp(k1 - km1), r2b = kan_rcp(k2 - k0);
  const float r3a = kan_rcp(k1 - km2), r3b = kan_rcp(k2 - km1), r3c = kan_rcp(k3 - k0);
  const float b1a = (k1 - x) * r1, b1b = (x - k0) * r1;
  const float u_a = b1a * r2a, u_b = b1b * r2b;
  const float b2a = (k1 - x) * u_a, b2b = (x - km1) * u_a + (k2 - x) * u_b, b2c = (x - k0) * u_b;
  const float q_a = b2a * r3a, q_b = b2b * r3b, q_c = b2c * r3c;
  const float w_a = -2.0f * u_a * r3a, w_b = 2.0f * (u_a - u_b) * r3b, w_c = 2.0f * u_b * r3c;
  const bool in = out.inside != 0;            // selects, not products: NaN / inf must not leak from outside the grid
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
  if (kNeedD3) {
    const float z_a = 2.0f * r1 * r2a * r3a, z_b = -2.0f * r1 * (r2a + r2b) * r3b, z_c = 2.0f * r1 * r2b * r3c;
    out.d3[0] = in ? -3.0f * z_a : 0.0f;
    out.d3[1] = in ? 3.0f * (z_a - z_b) : 0.0f;
    out.d3[2] = in ? 3.0f * (z_b - z_c) : 0.0f;
    out.d3[3] = in ? 3.0f * z_c : 0.0f;
  } else {
    out.d3[0] = out.d3[1] = out.d3[2] = out.d3[3] = 0.0f;
  }
}

// 1/h when the nk knots are equally spaced to within 1e-6*h, else 0.  The default
// grids of grid_size 5 and 10 are (fp32 rounding of the knot values moves their
// spacing by <= 4.5e-7*h); at grid_size 20 the rounding is 1.5e-6*h, which would
// cost ~8e-6 in the third derivative, so those grids keep the general path.
template <typename KnotPtr>
KAN_HD float equal_spacing_inv_h(const KnotPtr g, int nk) {
  const float h = (g[nk - 1] - g[0]) / (float)(nk - 1);
  float dev = 0.0f;
  for (int k = 0; k + 1 < nk; ++k) dev = fmaxf(dev, fabsf((g[k + 1] - g[k]) - h));
  return (h > 0.0f && dev <= 1e-6f * h) ? 1.0f / h : 0.0f;
}

// What the kernels know about a feature's knot row, in one float:
//    1/h   equally spaced (see above): the fixed polynomial pieces apply;
//   -1/h   not equally spaced, but increasing with every knot within h/4 of g_0 + k*h, h = (g_last - g_0)/(nk-1):
//          floor((x - g_0)/h) is then at most one interval away from the interval that holds x
//          (eval_basis_near_uniform); the default grid of grid_size 20 is of this kind;
//    0     anything else (a grid moved by update_grid): interval search.
template <typename KnotPtr>
KAN_HD float knot_spacing_info(const KnotPtr g, int nk) {
  const float eq = equal_spacing_inv_h(g, nk);
  if (eq > 0.0f) return eq;
  const float g0 = g[0], h = (g[nk - 1] - g0) / (float)(nk - 1);
  if (!(h > 0.0f)) return 0.0f;
  bool ok = true;
  for (int k = 0; k < nk; ++k) {
    ok = ok && fabsf(g[k] - (g0 + (float)k * h)) <= 0.25f * h;
    if (k + 1 < nk) ok = ok && (g[k + 1] > g[k]);
  }
  return ok ? -1.0f / h : 0.0f;
}

// Knot rows staged in LDS carry one extra float after the nk knots: 1/h when the
// feature's knots are equally spaced, 0 or negative otherwise (knot_spacing_info).
template <bool kNeedD3, typename KnotPtr>
KAN_HD void eval_basis_row(const KnotPtr row, int nk, float x, Basis& out) {
  const float inv_h = row[nk];
  if (inv_h > 0.0f) eval_basis_uniform<kNeedD3>(row, nk, inv_h, x, out);
  else eval_basis<kNeedD3>(row, nk, x, out);
}

// Base activation and its first three derivatives (kan.py:158; the scripts
// pass nn.Identity, the module default is SiLU).
struct Act { float f0, f1, f2, f3; };

KAN_HD Act eval_act(int kind, float x) {
  Act a;
  if (kind == kActSiLU) {
    const float s = kan_rcp(1.0f + kan_exp(-x));
    const float sp = s * (1.0f - s);        // sigmoid'
    const float t = 1.0f - 2.0f * s;
    a.f0 = x * s;
    a.f1 = s * (1.0f + x * (1.0f - s));
    a.f2 = sp * (2.0f + x * t);
    a.f3 = sp * (t * (3.0f + x * t) - 2.0f * x * sp);
  } else {
    a.f0 = x; a.f1 = 1.0f; a.f2 = 0.0f; a.f3 = 0.0f;
  }
  return a;
}

// Jet layout of one scalar quantity carried through the network, J = 1+D+D2
// floats:  [0] value, [1..D] d/dx_d, [1+D..D+D2] d2/dx_d^2 for d < D2.
template <int D, int D2>
struct JetShape {
  static constexpr int kD = D, kD2 = D2, kJ = 1 + D + D2;
};

// Coefficients that turn (phi, phi', phi'') of one edge function into the
// output-jet contribution of that edge, given the input jet a[]:
//   y[0]      += phi
//   y[1+d]    += phi'  * a[1+d]
//   y[1+D+d]  += phi'' * a[1+d]^2 + phi' * a[1+D+d]
template <int D, int D2>
KAN_HD void jet_accumulate(float* y, const float* a, float p0, float p1, float p2) {
  y[0] += p0;
#pragma unroll
  for (int d = 0; d < D; ++d) y[1 + d] += p1 * a[1 + d];
#pragma unroll
  for (int d = 0; d < D2; ++d) y[1 + D + d] += p2 * a[1 + d] * a[1 + d] + p1 * a[1 + D + d];
}

// The three scalars that pair with (phi, phi', phi'') when a cotangent gy[] of
// the output jet is pulled back onto one edge:  <gy, dy> = A0*phi + A1*phi' + A2*phi''.
template <int D, int D2>
KAN_HD void jet_pullback_coeffs(const float* gy, const float* a, float& A0, float& A1, float& A2) {
  A0 = gy[0];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int d = 0; d < D; ++d) s1 += gy[1 + d] * a[1 + d];
#pragma unroll
  for (int d = 0; d < D2; ++d) {
    s1 += gy[1 + D + d] * a[1 + D + d];
    s2 += gy[1 + D + d] * a[1 + d] * a[1 + d];
  }
  A1 = s1; A2 = s2;
}

#ifdef __HIPCC__
// Block-uniform: every feature's knot row is equally spaced or nearly so (ginfo != 0, see knot_spacing_info): the
// condition under which the fp16 matrix kernels of kan_f16.hip run; kernels that work beside them then leave the layer to them.
__device__ __forceinline__ bool all_rows_near_uniform(const float* ginfo, int I) {
  if (!ginfo) return false;
  const int lane = threadIdx.x & 63;
  bool near = true;
  for (int i = lane; i < ((I + 63) & ~63); i += 64)
    near = near && (__builtin_amdgcn_ballot_w64(ginfo[i < I ? i : 0] != 0.0f) == ~0ull);
  return near;
}
#endif

}  // namespace pdekan
