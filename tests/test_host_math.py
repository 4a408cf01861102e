"""kan_math.h (the inline math every kernel calls) compiled for the host and
checked against the oracle and the golden fixtures.  No GPU needed."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_state_dict, max_rel
from oracle import kan_oracle as O

SIGMA, RATE = O.BS_SIGMA, O.BS_RATE
FP = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="session")
def hm():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libhostmath.so")
    src = os.path.join(ROOT, "tests", "host_math.cpp")
    hdr = os.path.join(ROOT, "pde_amd", "csrc", "kan_math.h")
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", src, "-o", lib])
    return ctypes.CDLL(lib)


def fptr(a):
    return a.ctypes.data_as(FP)


def dense_from_local(hm, knots, x, C):
    """Scatter the 4 live bases (and derivatives) into dense [4][C] rows."""
    out = np.zeros(16, np.float32)
    c0, inside = ctypes.c_int(), ctypes.c_int()
    hm.hm_eval_basis(fptr(knots), len(knots), ctypes.c_float(x), fptr(out), ctypes.byref(c0), ctypes.byref(inside))
    dense = np.zeros((4, C), np.float64)
    for m in range(4):
        c = c0.value + m
        if 0 <= c < C:
            dense[:, c] = out[m::4]
    return dense, inside.value


def dense_reference(knots, xs, C, orders=4):
    """Dense Cox-de Boor in fp64 and its autograd derivatives: [orders][N][C]."""
    g64 = torch.from_numpy(np.asarray(knots)).double().unsqueeze(0)
    xt = torch.from_numpy(np.asarray(xs, np.float32)).double().unsqueeze(1).requires_grad_(True)
    cur = [O.b_splines(xt, g64, 3)[:, 0, c] for c in range(C)]
    out = [torch.stack(cur, 1)]
    for _ in range(orders - 1):
        cur = [torch.autograd.grad(t.sum(), xt, create_graph=True)[0][:, 0] for t in cur]
        out.append(torch.stack(cur, 1))
    return torch.stack(out).detach().numpy()


@pytest.mark.parametrize("G", [5, 10, 20])
def test_local_basis_matches_dense_cox_de_boor(hm, G):
    knots = O.uniform_grid(1, G)[0].numpy().copy()
    C = G + 3
    h = 2.0 / G
    xs = np.concatenate([np.linspace(-1.0 - 3 * h - 0.05, 1.0 + 3 * h + 0.05, 301),
                         knots.astype(np.float64), [-7.0, 9.0]]).astype(np.float32)
    ref = dense_reference(knots, xs, C)
    scale = np.array([1.0, 1 / h, 1 / h ** 2, 1 / h ** 3])[:, None]
    for n, x in enumerate(xs):
        dense, inside = dense_from_local(hm, knots, float(x), C)
        err = np.abs((dense - ref[:, n]) / scale).max()
        assert err < 5e-6, (float(x), err)
        assert inside == int(knots[0] <= x < knots[-1])


def test_equal_spacing_rule(hm):
    """Which grids take the polynomial fast path: the default grids of size 5 and
    10 do, size 20 (knot rounding ~2.5e-6*h) and any moved grid do not."""
    hm.hm_equal_spacing_inv_h.restype = ctypes.c_float
    for G, expect in ((5, True), (10, True), (20, False)):
        knots = O.uniform_grid(1, G)[0].numpy().copy()
        inv_h = hm.hm_equal_spacing_inv_h(fptr(knots), len(knots))
        assert (inv_h > 0) == expect, (G, inv_h)
        if expect:
            assert abs(inv_h - G / 2.0) < 1e-4
    moved = O.uniform_grid(1, 10)[0].numpy().copy()
    moved[7] += 1e-4
    assert hm.hm_equal_spacing_inv_h(fptr(moved), len(moved)) == 0.0


@pytest.mark.parametrize("G", [5, 10])
def test_equal_spacing_fast_path_matches_dense_cox_de_boor(hm, G):
    """The polynomial fast path the kernels take on the reference's default grid."""
    knots = O.uniform_grid(1, G)[0].numpy().copy()
    C = G + 3
    h = 2.0 / G
    xs = np.concatenate([np.linspace(-1.0 - 3 * h - 0.05, 1.0 + 3 * h + 0.05, 401),
                         knots.astype(np.float64), [-7.0, 9.0]]).astype(np.float32)
    ref = dense_reference(knots, xs, C)
    scale = np.array([1.0, 1 / h, 1 / h ** 2, 1 / h ** 3])[:, None]
    out = np.zeros(16, np.float32)
    c0, inside = ctypes.c_int(), ctypes.c_int()
    worst = 0.0
    for n, x in enumerate(xs):
        hm.hm_eval_basis_row(fptr(knots), len(knots), 1, ctypes.c_float(x), fptr(out), ctypes.byref(c0),
                             ctypes.byref(inside))
        dense = np.zeros((4, C))
        for m in range(4):
            c = c0.value + m
            if 0 <= c < C:
                dense[:, c] = out[m::4]
        worst = max(worst, np.abs((dense - ref[:, n]) / scale).max())
        assert inside.value == int(knots[0] <= x < knots[-1])
    assert worst < 5e-6, worst


@pytest.mark.parametrize("G", [5, 10])
def test_loop_free_equal_spacing_path_is_the_same_function(hm, G):
    """eval_basis_uniform_straight (selects instead of loops) against eval_basis_uniform: bit-identical,
    including at the knots, outside the grid and for non-finite input."""
    knots = O.uniform_grid(1, G)[0].numpy().copy()
    h = 2.0 / G
    around = np.concatenate([np.nextafter(knots, -np.inf), knots, np.nextafter(knots, np.inf)])
    xs = np.concatenate([np.linspace(-1.0 - 3 * h - 0.05, 1.0 + 3 * h + 0.05, 2001), around,
                         [-7.0, 9.0, 1e30, -1e30, np.inf, -np.inf]]).astype(np.float32)
    a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
    ca, ia, cb, ib = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    for x in xs:
        hm.hm_eval_basis_row(fptr(knots), len(knots), 1, ctypes.c_float(x), fptr(a), ctypes.byref(ca), ctypes.byref(ia))
        hm.hm_eval_basis_uniform_straight(fptr(knots), len(knots), ctypes.c_float(x), fptr(b), ctypes.byref(cb),
                                          ctypes.byref(ib))
        assert ia.value == ib.value
        if ia.value:
            assert ca.value == cb.value, x
        assert np.array_equal(a, b), x
        # and the variant that reads the three candidate knots in one round (fp16 matrix input-gradient kernel)
        hm.hm_eval_basis_uniform_oneshot(fptr(knots), len(knots), ctypes.c_float(x), fptr(b), ctypes.byref(cb),
                                         ctypes.byref(ib))
        assert ia.value == ib.value
        if ia.value:
            assert ca.value == cb.value, x
        assert np.array_equal(a, b), x


@pytest.mark.parametrize("G", [5, 10])
def test_quotient_interval_path_matches_up_to_the_second_derivative(hm, G):
    """eval_basis_uniform_by_quotient picks the interval without comparing against the knots; at a knot tie it may
    sit one interval off and evaluate the neighbouring piece a rounding error outside [0,1).  B, B', B'' (dense
    over the columns) must still agree with the exact-interval path everywhere, ties included: to fp32 rounding
    away from the knots, and at a tie to the mismatch between the stored knot spacing and 1/h (the grid counts as
    equally spaced up to 1e-6 h), which moves t by that much across the joint."""
    knots = O.uniform_grid(1, G)[0].numpy().copy()
    C = G + 3
    h = 2.0 / G
    around = np.concatenate([np.nextafter(knots, -np.inf), knots, np.nextafter(knots, np.inf)])
    xs = np.concatenate([np.linspace(-1.0 - 3 * h - 0.05, 1.0 + 3 * h + 0.05, 2001), around,
                         [-7.0, 9.0, 1e30, -1e30]]).astype(np.float32)
    a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
    ca, ia, cb, ib = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    scale = np.array([1.0, 1 / h, 1 / h ** 2])[:, None]
    worst = 0.0
    for x in xs:
        hm.hm_eval_basis_row(fptr(knots), len(knots), 1, ctypes.c_float(x), fptr(a), ctypes.byref(ca), ctypes.byref(ia))
        hm.hm_eval_basis_uniform_by_quotient(fptr(knots), len(knots), ctypes.c_float(x), fptr(b), ctypes.byref(cb),
                                             ctypes.byref(ib))
        assert ia.value == ib.value
        da, db = np.zeros((3, C)), np.zeros((3, C))
        for m in range(4):
            for dense, out, c0 in ((da, a, ca.value), (db, b, cb.value)):
                if 0 <= c0 + m < C:
                    dense[:, c0 + m] = out[m:12:4]
        worst = max(worst, np.abs((da - db) / scale).max())
    assert worst < 4e-6, worst


def test_knot_spacing_classes(hm):
    """1/h for equally spaced rows, -1/h for rows within h/4 of equal spacing (the default grid of size 20, a
    jittered grid), 0 for anything further off or not increasing."""
    hm.hm_knot_spacing_info.restype = ctypes.c_float
    for G, sign in ((5, 1), (10, 1), (20, -1)):
        knots = O.uniform_grid(1, G)[0].numpy().copy()
        v = hm.hm_knot_spacing_info(fptr(knots), len(knots))
        assert np.sign(v) == sign and abs(abs(v) - G / 2.0) < 1e-4, (G, v)
    rng = np.random.default_rng(3)
    knots = O.uniform_grid(1, 10)[0].numpy().copy()
    jit = (knots + rng.uniform(-0.2, 0.2, len(knots)) * 0.2).astype(np.float32)
    jit[0], jit[-1] = knots[0], knots[-1]
    assert hm.hm_knot_spacing_info(fptr(jit), len(jit)) < 0
    far = knots.copy(); far[7] += 0.3 * 0.2
    assert hm.hm_knot_spacing_info(fptr(far), len(far)) == 0.0
    tied = knots.copy(); tied[7] = tied[8]
    assert hm.hm_knot_spacing_info(fptr(tied), len(tied)) == 0.0


@pytest.mark.parametrize("kind", ["default20", "default10", "jittered"])
def test_near_uniform_path_matches_dense_cox_de_boor(hm, kind):
    """eval_basis_near_uniform (8 knots around the quotient guess, interval settled in registers, general triangle)
    against dense Cox-de Boor in fp64 on the same fp32 knots: B and its first three derivatives everywhere, knots and
    their fp32 neighbours included, and the reference's half-open interval convention (the third derivative jumps at
    the knots: the interval must be the exact one)."""
    G = 20 if kind == "default20" else 10
    knots = O.uniform_grid(1, G)[0].numpy().copy()
    h = 2.0 / G
    if kind == "jittered":
        rng = np.random.default_rng(5)
        inner = knots[1:-1] + (rng.uniform(-0.2, 0.2, len(knots) - 2) * h).astype(np.float32)
        knots = np.concatenate([knots[:1], inner, knots[-1:]]).astype(np.float32)
    C = G + 3
    around = np.concatenate([np.nextafter(knots, -np.inf), knots, np.nextafter(knots, np.inf)])
    xs = np.concatenate([np.linspace(-1.0 - 3 * h - 0.05, 1.0 + 3 * h + 0.05, 3001), around,
                         [-7.0, 9.0, 1e30, -1e30, np.inf, -np.inf, np.nan]]).astype(np.float32)
    ref = dense_reference(knots, xs, C, orders=4)
    hmin = np.diff(knots.astype(np.float64)).min()
    scale = np.array([1.0, 1 / hmin, 1 / hmin ** 2, 1 / hmin ** 3])[:, None]
    b = np.zeros(16, np.float32)
    cb, ib = ctypes.c_int(), ctypes.c_int()
    worst = 0.0
    for n, x in enumerate(xs):
        hm.hm_eval_basis_near_uniform(fptr(knots), len(knots), ctypes.c_float(x), fptr(b), ctypes.byref(cb),
                                      ctypes.byref(ib))
        assert ib.value == int(knots[0] <= x < knots[-1])
        if not ib.value:
            assert not b.any()
            continue
        s = cb.value + 3
        assert knots[s] <= x < knots[s + 1]
        dense = np.zeros((4, C))
        for m in range(4):
            if 0 <= cb.value + m < C:
                dense[:, cb.value + m] = b[m:16:4]
        worst = max(worst, np.abs((dense - ref[:4, n]) / scale).max())
    assert worst < (4e-6 if kind != "jittered" else 1.5e-5), worst


def test_local_basis_nonuniform_knots(hm):
    rng = np.random.default_rng(0)
    knots = np.sort(rng.uniform(-2, 2, 12)).astype(np.float32)
    knots += np.arange(12, dtype=np.float32) * 0.05          # keep them apart
    xs = rng.uniform(knots[0] - 0.2, knots[-1] + 0.2, 200).astype(np.float32)
    ref = dense_reference(knots, xs, 8, orders=3)
    for n, x in enumerate(xs):
        dense, _ = dense_from_local(hm, knots, float(x), 8)
        for k in range(3):
            assert np.abs(dense[k] - ref[k, n]).max() < 1e-5 * max(1.0, np.abs(ref[k]).max())


def test_silu_derivatives(hm):
    out = np.zeros(4, np.float32)
    for x in np.linspace(-6, 6, 49):
        hm.hm_eval_act(1, ctypes.c_float(x), fptr(out))
        xt = torch.tensor(float(np.float32(x)), dtype=torch.float64, requires_grad=True)
        f0 = torch.nn.functional.silu(xt)
        f1, = torch.autograd.grad(f0, xt, create_graph=True)
        f2, = torch.autograd.grad(f1, xt, create_graph=True)
        f3, = torch.autograd.grad(f2, xt)
        ref = np.array([f0.item(), f1.item(), f2.item(), f3.item()])
        assert np.abs(out - ref).max() < 2e-6
    hm.hm_eval_act(0, ctypes.c_float(0.3), fptr(out))
    assert np.allclose(out, [0.3, 1, 0, 0])


def pack(layer):
    sw = layer["spline_weight"].numpy()
    if layer.get("spline_scaler") is not None:
        sw = sw * layer["spline_scaler"].numpy()[..., None]
    return np.ascontiguousarray(np.concatenate([sw, layer["base_weight"].numpy()[..., None]], -1), np.float32)


def host_step(hm, layers, x, act):
    """The train step as the kernels compute it, on the host."""
    N, d = x.shape
    D, D2 = d, d - 1
    J = 1 + D + D2
    actk = {"identity": 0, "silu": 1}[act]
    jets = np.zeros((N, d, J), np.float32)
    jets[:, :, 0] = x
    for i in range(d):
        jets[:, i, 1 + i] = 1.0
    acts, packed = [jets], [pack(l) for l in layers]
    for l, wp in zip(layers, packed):
        Oo, I = l["base_weight"].shape
        G = l["spline_weight"].shape[2] - 3
        y = np.zeros((N, Oo, J), np.float32)
        grid = np.ascontiguousarray(l["grid"].numpy(), np.float32)
        assert hm.hm_layer_fwd(D, D2, fptr(acts[-1]), fptr(grid), fptr(wp), fptr(y), ctypes.c_int64(N), I, Oo, G, actk) == 0
        acts.append(y)
    v = acts[-1][:, 0, :].astype(np.float64)
    res = v[:, D] + RATE * v[:, 0]
    for j in range(D2):
        res = res - 0.5 * SIGMA ** 2 * x[:, j].astype(np.float64) ** 2 * v[:, 1 + D + j] - RATE * x[:, j] * v[:, 1 + j]
    loss = (res ** 2).mean()
    g = 2.0 * res / N
    gy = np.zeros((N, 1, J), np.float32)
    gy[:, 0, 0] = g * RATE
    gy[:, 0, D] = g
    for j in range(D2):
        gy[:, 0, 1 + j] = -g * RATE * x[:, j]
        gy[:, 0, 1 + D + j] = -g * 0.5 * SIGMA ** 2 * x[:, j].astype(np.float64) ** 2
    grads = [None] * len(layers)
    for li in range(len(layers) - 1, -1, -1):
        l, wp = layers[li], packed[li]
        Oo, I = l["base_weight"].shape
        G = l["spline_weight"].shape[2] - 3
        ga = np.zeros((N, I, J), np.float32)
        gw = np.zeros_like(wp)
        grid = np.ascontiguousarray(l["grid"].numpy(), np.float32)
        assert hm.hm_layer_bwd(D, D2, fptr(acts[li]), fptr(gy), fptr(grid), fptr(wp), fptr(ga), fptr(gw),
                               ctypes.c_int64(N), I, Oo, G, actk) == 0
        sw, sc = l["spline_weight"].numpy(), l["spline_scaler"].numpy()
        grads[li] = {"base_weight": gw[..., -1], "spline_weight": gw[..., :-1] * sc[..., None],
                     "spline_scaler": (gw[..., :-1] * sw).sum(-1)}
        gy = ga
    return {"v": v[:, 0], "first": v[:, 1:1 + D], "second": v[:, 1 + D:], "res": res, "loss": loss, "grads": grads}


@pytest.mark.parametrize("case", ["cfg1_silu", "cfg1_identity", "cfg2_silu", "cfg4_narrow", "cfg5_narrow",
                                  "updated_grid", "deeponet_trunk"])
def test_host_step_matches_reference(hm, golden, case):
    rec = golden(case)
    layers = O.layers_from_state_dict(golden_state_dict(rec))
    out = host_step(hm, layers, rec["x"], str(rec["act"]))
    tol = 1e-5
    assert max_rel(out["v"], rec["v64"]) <= tol
    assert max_rel(out["first"], rec["first64"]) <= tol
    assert max_rel(out["second"], rec["second64"]) <= tol
    assert max_rel(out["res"], rec["res64"]) <= tol
    assert max_rel(out["loss"], rec["loss64"]) <= tol
    for l, g in enumerate(out["grads"]):
        for key, val in g.items():
            r64, r32 = rec[f"grad64.layers.{l}.{key}"], rec[f"grad.layers.{l}.{key}"]
            assert max_rel(val, r64) <= tol, (l, key)
            # against the fp32 reference allow that reference's own fp32 noise on top
            assert max_rel(val, r32) <= tol + max_rel(r32, r64), (l, key)
