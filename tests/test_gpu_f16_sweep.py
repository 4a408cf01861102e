"""Seeded sweep over layer shapes, point counts, activations, knot rows and jet counts: the fp16 matrix kernels (and
the kernels that run beside them for the points and knot rows they leave out) against the fp32 kernels, one layer at
a time through the layer-level entry points.  The fixed cases of test_gpu_f16.py pin the block / pass decodes; this
one is for the combinations nobody thought of."""
import random

import pytest
import torch

from conftest import max_rel

pytestmark = pytest.mark.gpu


def _case(rng):
    jets = rng.choice([4, 4, 12])
    if jets == 4:
        G = rng.choice([10, 10, 20])
        I = 8 * rng.randint(1, 32 if G == 10 else 24)
        O_ = 64 * rng.randint(1, 4)
    else:
        G = 10
        I = 32 * rng.randint(1, 6)
        O_ = 64 * rng.choice([1, 2, 4])
    N = rng.choice([rng.randint(1, 40), rng.randint(41, 700), rng.randint(701, 5000)])
    act = rng.choice(["silu", "identity"])
    knots = rng.choice(["default", "default", "jittered", "moved"])
    return jets, I, O_, G, N, act, knots


CASES = [_case(random.Random(1000 + i)) for i in range(96)]


@pytest.mark.parametrize("jets,I,O_,G,N,act,knots", CASES)
def test_random_layers_agree_with_the_fp32_kernels(jets, I, O_, G, N, act, knots, monkeypatch):
    from pde_amd.kan import KANLinear
    torch.manual_seed(I * 7 + O_ + N)
    lay = KANLinear(I, O_, grid_size=G, base_activation=torch.nn.SiLU if act == "silu" else torch.nn.Identity).cuda()
    with torch.no_grad():
        for p in lay.parameters():
            p.add_(0.1 * torch.randn_like(p))
        if knots == "jittered":
            lay.grid[:, 5] += 0.2 * 0.04 * (10 / G)
            lay.grid[::3, 9] -= 0.2 * 0.02 * (10 / G)
        if knots == "moved":
            lay.grid[:, 6] += 0.45 * 2.0 / G
    g = torch.Generator().manual_seed(N)
    xj = torch.randn(N, I, jets, generator=g)
    xj[:, :, 0] = torch.rand(N, I, generator=g) * 3.6 - 1.8
    gy = torch.randn(N, O_, jets, generator=g)
    xj, gy = xj.cuda(), gy.cuda()
    nd, ns = (2, 1) if jets == 4 else (6, 5)

    def run():
        for p in lay.parameters():
            p.grad = None
        x = xj.clone().requires_grad_(True)
        y = lay.forward_jet(x, nd, ns)
        y.backward(gy)
        return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in lay.parameters()]

    a = run()
    monkeypatch.setenv("PDEKAN_DISABLE_F16", "1")
    b = run()
    monkeypatch.delenv("PDEKAN_DISABLE_F16")
    for name, u, v in zip(["y", "gx", "base_weight", "spline_weight", "spline_scaler"], a, b):
        assert torch.isfinite(u).all(), name
        assert max_rel(u.cpu(), v.cpu()) <= 6e-6, (name, max_rel(u.cpu(), v.cpu()))


def _model_case(rng):
    d = rng.choice([2, 2, 6])
    G = rng.choice([5, 10, 10, 20])
    hidden = [rng.choice([8, 16, 32, 64, 64, 96, 128]) for _ in range(rng.randint(1, 3))]
    N = rng.choice([rng.randint(1, 64), rng.randint(65, 1500)])
    mb = rng.choice([0, 0, 256, 1000])
    act = rng.choice(["silu", "identity"])
    return [d] + hidden + [1], G, N, mb, act


MODEL_CASES = [_model_case(random.Random(5000 + i)) for i in range(24)]


@pytest.mark.parametrize("widths,G,N,mb,act", MODEL_CASES)
def test_random_models_step_the_same_with_and_without_the_fp16_kernels(widths, G, N, mb, act, monkeypatch):
    """The fused step over random models (2 or 6 inputs, 1 to 3 hidden layers of mixed widths, micro-batches): whatever
    mix of matrix, sparse, split-pass and streaming kernels the layers get, loss and gradients agree with the run that
    keeps the fp16 kernels out, and with the fp64 oracle.  Points within 1e-6 of a knot are left out (there the side
    of the third derivative's jump depends on the last bit of the forward pass)."""
    from oracle import kan_oracle as O
    from pde_amd.pinn import BlackScholesStep
    from test_gpu_parity import _points_clear_of_knots, _random_model
    m = _random_model(widths, G, 31 + N, act=torch.nn.SiLU if act == "silu" else torch.nn.Identity)
    x = _points_clear_of_knots(m, N, act, 17 + N)
    ref = O.pinn_step(x.double(), O.layers_from_state_dict(m.state_dict(), dtype=torch.float64), act)
    ref32 = O.pinn_step(x, O.layers_from_state_dict({k: v.clone() for k, v in m.state_dict().items()}), act)
    flat64 = torch.cat([ref["grads"][l][k].flatten() for l in range(len(widths) - 1)
                        for k in ("base_weight", "spline_weight", "spline_scaler")])
    flat32 = torch.cat([ref32["grads"][l][k].flatten() for l in range(len(widths) - 1)
                        for k in ("base_weight", "spline_weight", "spline_scaler")])
    mg = m.cuda()
    a = BlackScholesStep(mg, micro_batch=mb)
    loss_a = a(x.cuda()).item()
    ga = a.flat_grad.clone()
    monkeypatch.setenv("PDEKAN_DISABLE_F16", "1")
    b = BlackScholesStep(mg, micro_batch=mb)
    loss_b = b(x.cuda()).item()
    monkeypatch.delenv("PDEKAN_DISABLE_F16")
    assert abs(loss_a - loss_b) <= 1e-5 * abs(loss_b)
    # the fp32 bar, plus what the reference algorithm itself loses in fp32 on this model
    bar = 1e-5 + max_rel(flat32, flat64)
    assert abs(loss_a - ref["loss"].item()) <= bar * abs(ref["loss"].item())
    assert max_rel(ga.cpu(), flat64) <= bar, max_rel(ga.cpu(), flat64)
    assert max_rel(b.flat_grad.cpu(), flat64) <= bar
