import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """A fixture file as a dict of numpy arrays (no pickles inside)."""
    with np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_state_dict(rec, prefix="sd.", dtype=torch.float32):
    return {k[len(prefix):]: torch.from_numpy(v).to(dtype) for k, v in rec.items() if k.startswith(prefix)}


def max_rel(a, b):
    """max|a-b| / max|b| per tensor: the parity metric used throughout."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    denom = b.abs().max().item()
    if denom == 0.0:
        return (a - b).abs().max().item()
    return (a - b).abs().max().item() / denom


GOLDEN_CASES = ["cfg1_silu", "cfg1_identity", "cfg1_init", "cfg2_silu", "cfg2_identity",
                "cfg4_narrow", "cfg5_narrow", "updated_grid", "deeponet_trunk"]


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


DEEPONET_CASES = ["deeponet_run1", "deeponet_wide"]
DEEPONET_MODELS = ("model1", "model2", "model4", "model5")
