"""The CPU oracle against the fixtures generated from the reference itself."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, golden_state_dict, max_rel
from oracle import kan_oracle as O


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_oracle_matches_reference_fp32(golden, case):
    rec = golden(case)
    layers = O.layers_from_state_dict(golden_state_dict(rec))
    x = torch.from_numpy(rec["x"])
    out = O.pinn_step(x, layers, str(rec["act"]))
    # same ATen ops in the same order: the oracle should agree to the last bits
    assert max_rel(out["v"], rec["v"]) <= 1e-6
    assert max_rel(out["first"], rec["first"]) <= 1e-6
    assert max_rel(out["second"], rec["second"]) <= 1e-6
    assert max_rel(out["res"], rec["res"]) <= 1e-6
    assert max_rel(out["loss"], rec["loss"]) <= 1e-6
    for l, g in enumerate(out["grads"]):
        for key, val in g.items():
            assert max_rel(val, rec[f"grad.layers.{l}.{key}"]) <= 2e-6, (l, key)


@pytest.mark.parametrize("case", ["cfg1_silu", "cfg2_silu", "cfg5_narrow", "updated_grid"])
def test_oracle_matches_reference_fp64(golden, case):
    rec = golden(case)
    layers = O.layers_from_state_dict(golden_state_dict(rec), dtype=torch.float64)
    x = torch.from_numpy(rec["x"]).double()
    out = O.pinn_step(x, layers, str(rec["act"]))
    assert max_rel(out["res"], rec["res64"]) <= 1e-12
    assert max_rel(out["loss"], rec["loss64"]) <= 1e-12
    for l, g in enumerate(out["grads"]):
        for key, val in g.items():
            assert max_rel(val, rec[f"grad64.layers.{l}.{key}"]) <= 1e-11, (l, key)


@pytest.mark.parametrize("case", ["cfg1_silu", "cfg2_identity", "cfg5_narrow"])
def test_oracle_value_step(golden, case):
    rec = golden(case)
    layers = O.layers_from_state_dict(golden_state_dict(rec))
    x = torch.from_numpy(rec["x"])
    gy = torch.linspace(-1.0, 1.0, rec["y"].size).view(*rec["y"].shape)
    out = O.value_step(x, layers, str(rec["act"]), gy)
    assert max_rel(out["y"], rec["y"]) <= 1e-6
    assert max_rel(out["gx"], rec["gx"]) <= 1e-6
    for l, g in enumerate(out["grads"]):
        for key, val in g.items():
            assert max_rel(val, rec[f"vgrad.layers.{l}.{key}"]) <= 2e-6, (l, key)


def test_jacrev_hessian_formulation(golden):
    """run1.py:122-124 (jacrev/hessian then sum) equals per-point derivatives."""
    rec = golden("jacrev_hessian")
    layers = O.layers_from_state_dict(golden_state_dict(rec))
    x = torch.from_numpy(rec["x"]).requires_grad_(True)
    v, first, second = O.point_jets(x, layers, "identity", 1)
    res = O.bs_residual(x, v, first, second)
    assert max_rel(v, rec["s"]) <= 1e-6
    assert max_rel(first[:, 0], rec["s_x"]) <= 1e-6
    assert max_rel(first[:, 1], rec["s_t"]) <= 1e-6
    assert max_rel(second[:, 0], rec["s_xx"]) <= 1e-6
    assert max_rel(res, rec["res"]) <= 1e-6


def test_fp32_noise_floor_documented(golden):
    """fp32 vs fp64 of the reference itself: the floor the 1e-5 bar sits above."""
    rec = golden("cfg2_silu")
    assert max_rel(rec["res"], rec["res64"]) < 5e-6
    assert max_rel(rec["loss"], rec["loss64"]) < 5e-6


def test_uniform_grid_matches_reference_buffer(golden):
    rec = golden("cfg1_init")
    g = O.uniform_grid(2, 5)
    assert np.array_equal(g.numpy(), rec["init.layers.0.grid"])
