"""The DeepONet restatement (oracle/deeponet_oracle.py) against fixtures made by
the reference's own KAN class (tests/golden/deeponet_*.npz)."""
import pytest
import torch

from conftest import DEEPONET_CASES, DEEPONET_MODELS, golden_state_dict, max_rel
from oracle import deeponet_oracle as DO
from oracle import kan_oracle as O


def layer_sets(rec, dtype):
    return {m: O.layers_from_state_dict(golden_state_dict(rec, prefix=f"sd.{m}."), dtype=dtype)
            for m in DEEPONET_MODELS}


def batches(rec, dtype):
    t = lambda k: torch.from_numpy(rec[k]).to(dtype)
    return t("u1"), t("u2"), (t("x_i"), t("t_i"), t("out_i")), (t("x_b"), t("t_b"), t("out_b"))


@pytest.mark.parametrize("case", DEEPONET_CASES)
@pytest.mark.parametrize("dtype,tag,tol", [(torch.float64, "64", 1e-10), (torch.float32, "", 1e-5)])
def test_deeponet_oracle_matches_reference(golden, case, dtype, tag, tol):
    rec = golden(case)
    u1, u2, bc, pde = batches(rec, dtype)
    out = DO.step_with_grads(layer_sets(rec, dtype), "identity", u1, u2, bc, pde, float(rec["sigma"]),
                             float(rec["rate"]), float(rec["bc_weight"]))
    for key in ("s_pred", "res", "bc_loss", "pde_loss", "loss"):
        assert max_rel(out[key], rec[key + tag]) <= tol, key
    for name, g in out["grads"].items():
        assert max_rel(g, rec[f"grad{tag}.{name}"]) <= tol, name


def test_single_row_equals_repeated_rows(golden):
    """model4 on one row broadcast over the points equals the reference's n
    repeated rows (run1.py:112-113) - the identity pde_amd/deeponet.py relies on."""
    rec = golden("deeponet_run1")
    sets = layer_sets(rec, torch.float64)
    nets = DO.nets_from_layers(sets, "identity")
    u1, u2, bc, _ = batches(rec, torch.float64)
    x, t, _ = bc
    ref = DO.operator_net(nets, u1, u2, x, t)
    row = nets["model4"]((nets["model1"](u1) * nets["model2"](u2)).reshape(1, -1))
    mine = torch.sum(row * nets["model5"](DO.stack_xt(x, t)), dim=1)
    assert max_rel(mine, ref) <= 1e-14
