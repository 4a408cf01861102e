"""HIP path (through the C ABI) against the oracle and the golden fixtures.
Runs on the GPU box only: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, golden_state_dict, max_rel
from oracle import kan_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5   # max|a-b|/max|b| per tensor, fp32


def build_model(rec, device="cuda"):
    from pde_amd import KAN
    act = {"silu": torch.nn.SiLU, "identity": torch.nn.Identity}[str(rec["act"])]
    m = KAN([int(w) for w in rec["widths"]], grid_size=int(rec["grid_size"]), base_activation=act)
    m.load_state_dict(golden_state_dict(rec))
    return m.to(device)


def check(val, rec, key, tol=TOL):
    r32, r64 = rec[key], rec[key.replace(".", "64.", 1) if "." in key else key + "64"]
    v = val.detach().cpu().numpy() if torch.is_tensor(val) else val
    assert max_rel(v, r64) <= tol, (key, max_rel(v, r64))
    # the fp32 reference carries its own rounding noise; allow it on top
    assert max_rel(v, r32) <= tol + max_rel(r32, r64), (key, max_rel(v, r32))


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_fused_step_matches_reference(golden, case):
    from pde_amd.pinn import BlackScholesStep
    rec = golden(case)
    m = build_model(rec)
    x = torch.from_numpy(rec["x"]).cuda()
    step = BlackScholesStep(m)
    res = torch.empty(x.size(0), device="cuda")
    loss = step(x, residual_out=res)
    torch.cuda.synchronize()
    check(res, rec, "res")
    check(loss, rec, "loss")
    for name, p in m.named_parameters():
        check(p.grad, rec, f"grad.{name}")


@pytest.mark.parametrize("case", ["cfg1_silu", "cfg2_silu", "cfg5_narrow", "updated_grid"])
def test_fused_step_micro_batches_agree(golden, case):
    from pde_amd.pinn import BlackScholesStep
    rec = golden(case)
    m = build_model(rec)
    x = torch.from_numpy(rec["x"]).cuda()
    full = BlackScholesStep(m)
    l0 = full(x).item()
    g0 = full.flat_grad.clone()
    mb = BlackScholesStep(m, micro_batch=37)
    l1 = mb(x).item()
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    assert max_rel(mb.flat_grad.cpu(), g0.cpu()) <= 2e-6
    check(mb.flat_grad[: m.layers[0].base_weight.numel()].view_as(m.layers[0].base_weight), rec,
          "grad.layers.0.base_weight")


@pytest.mark.parametrize("case", ["cfg1_silu", "cfg1_identity", "cfg2_silu", "cfg5_narrow", "updated_grid"])
def test_forward_jet_matches_reference(golden, case):
    rec = golden(case)
    m = build_model(rec)
    x = torch.from_numpy(rec["x"]).cuda()
    d = x.size(1)
    jet = m.forward_jet(x)
    assert jet.shape == (x.size(0), 1, 2 * d)
    check(jet[:, 0, 0], rec, "v")
    check(jet[:, 0, 1:1 + d], rec, "first")
    check(jet[:, 0, 1 + d:], rec, "second")


@pytest.mark.parametrize("case", ["cfg1_silu", "cfg2_identity", "cfg4_narrow", "deeponet_trunk"])
def test_autograd_through_forward_jet(golden, case):
    """Residual + loss in tensor ops on top of forward_jet, backward through the
    jet kernels: must give the fused step's (= the reference's) gradients."""
    from pde_amd.pinn import bs_residual_from_jet
    rec = golden(case)
    m = build_model(rec)
    x = torch.from_numpy(rec["x"]).cuda()
    res = bs_residual_from_jet(x, m.forward_jet(x))
    loss = (res ** 2).mean()
    loss.backward()
    check(res, rec, "res")
    check(loss, rec, "loss")
    for name, p in m.named_parameters():
        check(p.grad, rec, f"grad.{name}")


@pytest.mark.parametrize("case", ["cfg1_silu", "cfg2_identity", "cfg5_narrow", "updated_grid"])
def test_plain_forward_backward(golden, case):
    """KAN.forward + loss.backward(): the data-fit use of the layer (run1.py:134-141)."""
    rec = golden(case)
    m = build_model(rec)
    x = torch.from_numpy(rec["x"]).cuda().requires_grad_(True)
    y = m(x)
    gy = torch.linspace(-1.0, 1.0, y.numel(), device="cuda").view_as(y)
    (y * gy).sum().backward()
    check(y, rec, "y")
    check(x.grad, rec, "gx")
    for name, p in m.named_parameters():
        check(p.grad, rec, f"vgrad.{name}")


def test_noncontiguous_and_batched_inputs(golden):
    """The scripts feed a permuted stack (strides (1, n), run1.py:101-106) and
    KANLinear accepts any leading dims (kan.py:155-165)."""
    rec = golden("deeponet_trunk")
    m = build_model(rec)
    x = torch.from_numpy(rec["x"]).cuda()
    y_ref = m(x)
    xt = torch.stack([x[:, 0].clone(), x[:, 1].clone()]).permute(1, 0)
    assert not xt.is_contiguous()
    assert torch.equal(m(xt), y_ref)
    assert torch.equal(m.forward_jet(xt), m.forward_jet(x))
    n = (x.size(0) // 2) * 2
    y3 = m(x[:n].view(2, n // 2, 2))
    assert y3.shape == (2, n // 2, 1)
    assert torch.equal(y3.view(n, 1), y_ref[:n])
    with pytest.raises(AssertionError):
        m(torch.zeros(4, 3, device="cuda"))


def test_cpu_tensor_is_refused(golden):
    rec = golden("cfg1_silu")
    m = build_model(rec, device="cpu")
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        m(torch.zeros(4, 2))


def test_large_batch_properties():
    """Full-size behaviour (2^20 points, the benchmark's batch) through size-independent
    properties: the fused step equals the sum of its two halves (additivity of the loss
    and of the gradients), and a permutation of the points changes nothing beyond fp32
    reassociation."""
    from pde_amd import KAN
    from pde_amd.pinn import BlackScholesStep
    torch.manual_seed(0)
    m = KAN([2, 64, 64, 1], grid_size=10).cuda()
    g = torch.Generator().manual_seed(0)
    n = 1 << 20
    x = torch.rand(n, 2, generator=g).cuda()
    step = BlackScholesStep(m)
    l_all = step(x).item()
    g_all = step.flat_grad.clone()
    la = step(x[: n // 2], n_global=n).item()
    ga = step.flat_grad.clone()
    lb = step(x[n // 2:], n_global=n).item()
    gb = step.flat_grad.clone()
    assert abs(l_all - (la + lb)) <= 1e-5 * abs(l_all)
    assert max_rel((ga + gb).cpu(), g_all.cpu()) <= 1e-5
    perm = torch.randperm(n, generator=g).cuda()
    lp = step(x[perm]).item()
    assert abs(lp - l_all) <= 1e-5 * abs(l_all)
    assert max_rel(step.flat_grad.cpu(), g_all.cpu()) <= 1e-5
    # against the oracle on a slice the CPU finishes in seconds
    sub = x[:2048].cpu()
    layers = O.layers_from_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    ref = O.pinn_step(sub, layers, "silu")
    res = torch.empty(2048, device="cuda")
    l_sub = step(x[:2048], residual_out=res).item()
    assert max_rel(res.cpu(), ref["res"]) <= 1e-5
    assert abs(l_sub - ref["loss"].item()) <= 1e-5 * abs(ref["loss"].item())
    flat_ref = torch.cat([ref["grads"][l][k].flatten() for l in range(3)
                          for k in ("base_weight", "spline_weight", "spline_scaler")])
    assert max_rel(step.flat_grad.cpu(), flat_ref) <= 1e-5


def _random_model(widths, G, seed, act=torch.nn.SiLU):
    from pde_amd import KAN
    torch.manual_seed(seed)
    m = KAN(widths, grid_size=G, base_activation=act)
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.1 * torch.randn(p.shape, generator=gen))
    return m


@pytest.mark.parametrize("G,N", [(10, 1), (10, 63), (10, 65), (10, 1500), (5, 777), (20, 500)])
def test_mfma_hidden_layer_step_matches_oracle(G, N):
    """[2,64,64,1]: the 64x64 layer takes the MFMA kernels (forward, input
    gradients, weight gradients); ragged N exercises the tile tails and several
    point chunks."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model([2, 64, 64, 1], G, 11 + G)
    layers = O.layers_from_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    x = torch.rand(N, 2, generator=torch.Generator().manual_seed(5))
    ref = O.pinn_step(x, layers, "silu")
    ref64 = O.pinn_step(x.double(), O.layers_from_state_dict(m.state_dict(), dtype=torch.float64), "silu")
    m = m.cuda()
    step = BlackScholesStep(m)
    res = torch.empty(N, device="cuda")
    loss = step(x.cuda(), residual_out=res).item()
    flat32 = torch.cat([ref["grads"][l][k].flatten() for l in range(3)
                        for k in ("base_weight", "spline_weight", "spline_scaler")])
    flat64 = torch.cat([ref64["grads"][l][k].flatten() for l in range(3)
                        for k in ("base_weight", "spline_weight", "spline_scaler")])
    assert max_rel(res.cpu(), ref64["res"]) <= TOL
    assert abs(loss - ref64["loss"].item()) <= TOL * abs(ref64["loss"].item())
    assert max_rel(step.flat_grad.cpu(), flat64) <= TOL
    assert max_rel(step.flat_grad.cpu(), flat32) <= TOL + max_rel(flat32, flat64)
    # per-layer too: a wrong small tensor must not hide behind a large one
    off = 0
    for l in range(3):
        for k in ("base_weight", "spline_weight", "spline_scaler"):
            n = ref64["grads"][l][k].numel()
            assert max_rel(step.flat_grad[off:off + n].cpu(), ref64["grads"][l][k].flatten()) <= TOL, (l, k)
            off += n


def test_mfma_and_streaming_paths_agree(monkeypatch):
    from pde_amd.pinn import BlackScholesStep
    m = _random_model([2, 64, 64, 1], 10, 3).cuda()
    x = torch.rand(4099, 2, generator=torch.Generator().manual_seed(9)).cuda()
    step = BlackScholesStep(m)
    l_mfma = step(x).item()
    g_mfma = step.flat_grad.clone()
    monkeypatch.setenv("PDEKAN_DISABLE_MFMA", "1")
    step = BlackScholesStep(m)          # workspace size depends on the path
    l_gen = step(x).item()
    g_gen = step.flat_grad.clone()
    monkeypatch.delenv("PDEKAN_DISABLE_MFMA")
    assert abs(l_mfma - l_gen) <= 2e-6 * abs(l_gen)
    assert max_rel(g_mfma.cpu(), g_gen.cpu()) <= 5e-6


@pytest.mark.parametrize("case", ["cfg1_silu", "cfg2_silu", "cfg5_narrow"])
def test_fused_head_and_separate_kernels_agree(golden, case, monkeypatch):
    """The fused last-layer + residual + loss + backward kernel against the
    same step done with the separate layer kernels."""
    from pde_amd.pinn import BlackScholesStep
    rec = golden(case)
    m = build_model(rec)
    x = torch.from_numpy(rec["x"]).cuda()
    fused = BlackScholesStep(m)
    r0 = torch.empty(x.size(0), device="cuda")
    l0 = fused(x, residual_out=r0).item()
    g0 = fused.flat_grad.clone()
    monkeypatch.setenv("PDEKAN_DISABLE_FUSED_HEAD", "1")
    sep = BlackScholesStep(m)
    r1 = torch.empty(x.size(0), device="cuda")
    l1 = sep(x, residual_out=r1).item()
    monkeypatch.delenv("PDEKAN_DISABLE_FUSED_HEAD")
    assert abs(l0 - l1) <= 2e-6 * abs(l1)
    assert max_rel(r0.cpu(), r1.cpu()) <= 2e-6
    assert max_rel(g0.cpu(), sep.flat_grad.cpu()) <= 5e-6
    check(sep.flat_grad[: m.layers[0].base_weight.numel()].view_as(m.layers[0].base_weight), rec,
          "grad.layers.0.base_weight")


def test_single_layer_model_step():
    """KAN [2,1]: the head is the whole model."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model([2, 1], 5, 21)
    layers = O.layers_from_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    x = torch.rand(300, 2, generator=torch.Generator().manual_seed(2))
    ref = O.pinn_step(x.double(), O.layers_from_state_dict(m.state_dict(), dtype=torch.float64), "silu")
    m = m.cuda()
    step = BlackScholesStep(m)
    loss = step(x.cuda()).item()
    flat = torch.cat([ref["grads"][0][k].flatten() for k in ("base_weight", "spline_weight", "spline_scaler")])
    assert abs(loss - ref["loss"].item()) <= TOL * abs(ref["loss"].item())
    assert max_rel(step.flat_grad.cpu(), flat) <= TOL


@pytest.mark.parametrize("disable", ["PDEKAN_DISABLE_SPARSE", "PDEKAN_DISABLE_MFMA", None])
@pytest.mark.parametrize("widths,G,N", [([2, 64, 64, 1], 10, 333), ([2, 64, 128, 64, 1], 10, 200),
                                        ([2, 64, 64, 1], 5, 129), ([2, 6, 64, 4, 128, 1], 10, 77)])
def test_hidden_layer_paths_match_oracle(widths, G, N, disable, monkeypatch):
    """The same step through each hidden-layer path: sparse VALU forward/input
    gradients (default where eligible), dense MFMA only, streaming only."""
    from pde_amd.pinn import BlackScholesStep
    if disable:
        monkeypatch.setenv(disable, "1")
    m = _random_model(widths, G, 31 + G + len(widths))
    x = torch.rand(N, 2, generator=torch.Generator().manual_seed(6))
    ref = O.pinn_step(x.double(), O.layers_from_state_dict(m.state_dict(), dtype=torch.float64), "silu")
    m = m.cuda()
    step = BlackScholesStep(m)
    res = torch.empty(N, device="cuda")
    loss = step(x.cuda(), residual_out=res).item()
    assert max_rel(res.cpu(), ref["res"]) <= TOL
    assert abs(loss - ref["loss"].item()) <= TOL * abs(ref["loss"].item())
    off = 0
    for l in range(len(widths) - 1):
        for k in ("base_weight", "spline_weight", "spline_scaler"):
            n = ref["grads"][l][k].numel()
            assert max_rel(step.flat_grad[off:off + n].cpu(), ref["grads"][l][k].flatten()) <= TOL, (l, k)
            off += n
    # forward_jet and autograd through it use the layer-level entry points
    jet = m.forward_jet(x.cuda())
    assert max_rel(jet[:, 0, 0].cpu(), ref["v"]) <= TOL
    assert max_rel(jet[:, 0, 3].cpu(), ref["second"][:, 0]) <= TOL


def test_moved_grid_takes_the_general_basis_path():
    """After update_grid the knots are no longer equally spaced: the kernels must
    fall back from the polynomial fast path to the local de Boor triangle (hidden
    64x64 layer: sparse forward/input-gradient kernels and the MFMA weight-gradient kernel)."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model([2, 64, 64, 1], 10, 77).cuda()
    gen = torch.Generator().manual_seed(78)
    samples = (torch.rand(512, 2, generator=gen) * 1.2 - 0.1).cuda()
    with torch.no_grad():
        m(samples, update_grid=True)              # kan.py:274-279 with update_grid=True
    g1 = m.layers[1].grid[0]
    steps = g1[1:] - g1[:-1]
    assert (steps.max() - steps.min()).item() > 1e-3      # really non-uniform now
    x = torch.rand(300, 2, generator=gen)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    ref = O.pinn_step(x.double(), O.layers_from_state_dict(sd, dtype=torch.float64), "silu")
    step = BlackScholesStep(m)
    res = torch.empty(300, device="cuda")
    loss = step(x.cuda(), residual_out=res).item()
    assert max_rel(res.cpu(), ref["res"]) <= TOL
    assert abs(loss - ref["loss"].item()) <= TOL * abs(ref["loss"].item())
    off = 0
    for l in range(3):
        for k in ("base_weight", "spline_weight", "spline_scaler"):
            n = ref["grads"][l][k].numel()
            assert max_rel(step.flat_grad[off:off + n].cpu(), ref["grads"][l][k].flatten()) <= TOL, (l, k)
            off += n


def _points_clear_of_knots(m_cpu, n, act, seed, clearance=1e-6, device="cpu"):
    """Seeded points none of whose hidden activations sits within `clearance` of a knot of the next layer.
    The third derivative of a cubic B-spline jumps at the knots and enters the input gradients, so a value
    within fp32 rounding of a knot (it happens once in ~10^5 activations) makes the gradients of the layers
    below depend on which side the rounding falls: any fp32 implementation, the reference's included, can
    differ from the fp64 oracle there by the whole contribution of that point."""
    layers = [{k: (v.to(device) if v is not None else None) for k, v in lay.items()}
              for lay in O.layers_from_state_dict(m_cpu.state_dict(), dtype=torch.float64)]
    x = torch.rand(n + n // 8 + 16, m_cpu.layers[0].in_features, generator=torch.Generator().manual_seed(seed))
    clear = torch.ones(x.size(0), dtype=torch.bool)
    for s0 in range(0, x.size(0), 1 << 14):            # slabs: the [points, features, knots] differences are large
        h = x[s0:s0 + (1 << 14)].double().to(device)
        ok = torch.ones(h.size(0), dtype=torch.bool, device=device)
        for li in range(len(layers) - 1):
            h = O.layer_forward(h, layers[li], act)
            ok &= (h[:, :, None] - layers[li + 1]["grid"][None, :, :]).abs().amin(dim=(1, 2)) >= clearance
        clear[s0:s0 + (1 << 14)] = ok.cpu()
    x = x[clear]
    assert x.size(0) >= n
    return x[:n].contiguous()


def _grads_vs_oracle(m_cpu, x, act, widths, tol=TOL):
    from pde_amd.pinn import BlackScholesStep
    ref = O.pinn_step(x.double(), O.layers_from_state_dict(m_cpu.state_dict(), dtype=torch.float64), act)
    ref32 = O.pinn_step(x, O.layers_from_state_dict({k: v.clone() for k, v in m_cpu.state_dict().items()}), act)
    m = m_cpu.cuda()
    step = BlackScholesStep(m)
    loss = step(x.cuda()).item()
    assert abs(loss - ref["loss"].item()) <= tol * abs(ref["loss"].item())
    off = 0
    for l in range(len(widths) - 1):
        for k in ("base_weight", "spline_weight", "spline_scaler"):
            r64 = ref["grads"][l][k].flatten()
            n = r64.numel()
            # the fp32 bar, plus what the reference algorithm itself loses in fp32 on this model
            bar = tol + max_rel(ref32["grads"][l][k].flatten(), r64)
            assert max_rel(step.flat_grad[off:off + n].cpu(), r64) <= bar, (l, k)
            off += n
    return step


@pytest.mark.parametrize("act", ["silu", "identity"])
@pytest.mark.parametrize("widths,N", [([2, 64, 64, 1], 16), ([2, 64, 64, 1], 4112), ([2, 64, 128, 64, 1], 1040),
                                      ([2, 64, 64, 1], 4099)])
def test_split_bf16_weight_gradients_match_oracle(widths, N, act, monkeypatch):
    """Grid 10, equally spaced knots: the weight gradients of the wide hidden layers come from the split-bf16
    matrix kernel (three bf16 pieces per fp32 operand, six products); when N is not a multiple of 16 the last
    N % 16 points take the fp32 kernel as a chunk of their own.  Against the fp64 oracle at the fp32 bar, and
    against the fp32 matrix kernel on the same inputs."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model(widths, 10, 41 + N, act=torch.nn.SiLU if act == "silu" else torch.nn.Identity)
    x = _points_clear_of_knots(m, N, act, 8)
    step = _grads_vs_oracle(m, x, act, widths)
    g_split = step.flat_grad.clone()
    monkeypatch.setenv("PDEKAN_DISABLE_BF16X3", "1")
    step32 = BlackScholesStep(m.cuda())
    step32(x.cuda())
    monkeypatch.delenv("PDEKAN_DISABLE_BF16X3")
    assert max_rel(g_split.cpu(), step32.flat_grad.cpu()) <= 2e-6


@pytest.mark.parametrize("widths,G,N,shard", [([2, 64, 64, 1], 10, 1 << 20, 1 << 15),
                                              ([2, 256, 256, 256, 1], 20, 1 << 17, 1 << 12),
                                              ([6, 128, 128, 1], 10, 1 << 17, 1 << 12)])
def test_large_batch_gradients_match_fp64_oracle(widths, G, N, shard):
    """The benchmark configuration at its full size (2^20 points), the wide grid-20 model and the 5-asset basket
    model (12 jets per value: 12-jet fp16 matrix kernels for the hidden layer, streaming first layer) at 2^17: loss and every
    parameter gradient of the fused step against the fp64 oracle, which is evaluated here on the GPU shard by shard
    (the loss is a sum over points, pinn_step's n_global).  Long sums are where accumulation order and rounding
    show: the matrix instructions' accumulate drifts with the length of a run of steps, which the weight-gradient
    kernels bound (chunks of points, periodic vector adds)."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model(widths, G, 77)
    # (a hidden value within fp32 rounding of a knot moves the gradients of the layers below by that point's whole
    # contribution: such points are dropped, see _points_clear_of_knots)
    x = _points_clear_of_knots(m, N, "silu", 1234, device="cuda")
    layers64 = [{k: (v.double().cuda() if v is not None else None) for k, v in lay.items()}
                for lay in O.layers_from_state_dict(m.state_dict())]
    keys = ("base_weight", "spline_weight", "spline_scaler")
    loss64, flat64 = 0.0, None
    for s0 in range(0, N, shard):
        r = O.pinn_step(x[s0:s0 + shard].double().cuda(), layers64, "silu", n_global=N)
        loss64 += r["loss"].item()
        flat = torch.cat([r["grads"][l][k].flatten() for l in range(len(widths) - 1) for k in keys])
        flat64 = flat if flat64 is None else flat64 + flat
        del r
    mg = m.cuda()
    step = BlackScholesStep(mg)
    loss = step(x.cuda()).item()
    assert abs(loss - loss64) <= TOL * abs(loss64)
    off = 0
    for l, lay in enumerate(mg.layers):
        for k in keys:
            n = getattr(lay, k).numel()
            err = max_rel(step.flat_grad[off:off + n].double(), flat64[off:off + n])
            assert err <= TOL, (l, k, err)
            off += n


@pytest.mark.parametrize("widths,G", [([2, 256, 256, 256, 1], 20), ([6, 128, 128, 1], 10)])
def test_wide_models_at_their_full_point_count(widths, G):
    """The wide grid-20 model and the 5-asset basket model at the full 2^22 points of their benchmark
    configurations, through properties that do not need an oracle of that size:
      - additivity: the step on all 2^22 points (micro-batches of 2^20) equals the sum of 32 steps on shards of 2^17
        points with n_global = 2^22, loss and every gradient tensor;
      - a permutation of the points changes nothing beyond fp32 reassociation;
      - the first shard's share (n_global = 2^22) against the fp64 oracle.
    Long runs of matrix-instruction accumulates and many partial chunks are what only this size exercises."""
    from pde_amd.pinn import BlackScholesStep
    N, S, shard64 = 1 << 22, 1 << 17, 1 << 12
    keys = ("base_weight", "spline_weight", "spline_scaler")
    m = _random_model(widths, G, 91)
    x = _points_clear_of_knots(m, N, "silu", 4321, device="cuda").cuda()
    layers64 = [{k: (v.double().cuda() if v is not None else None) for k, v in lay.items()}
                for lay in O.layers_from_state_dict(m.state_dict())]
    loss64, flat64 = 0.0, None
    for s0 in range(0, S, shard64):
        r = O.pinn_step(x[s0:s0 + shard64].double(), layers64, "silu", n_global=N)
        loss64 += r["loss"].item()
        flat = torch.cat([r["grads"][l][k].flatten() for l in range(len(widths) - 1) for k in keys])
        flat64 = flat if flat64 is None else flat64 + flat
        del r
    mg = m.cuda()
    bounds, off = [], 0
    for l, lay in enumerate(mg.layers):
        for k in keys:
            bounds.append((l, k, off, off + getattr(lay, k).numel()))
            off += getattr(lay, k).numel()

    def agree(a, b, what):
        for l, k, lo, hi in bounds:
            err = max_rel(a[lo:hi].double(), b[lo:hi].double())
            assert err <= TOL, (what, l, k, err)

    full = BlackScholesStep(mg, micro_batch=1 << 20)
    l_all = full(x).item()
    g_all = full.flat_grad.clone()
    part = BlackScholesStep(mg)
    l_sum, g_sum = 0.0, torch.zeros_like(g_all, dtype=torch.float64)
    for s0 in range(0, N, S):
        l_sum += part(x[s0:s0 + S], n_global=N).item()
        g_sum += part.flat_grad.double()
        if s0 == 0:
            assert abs(l_sum - loss64) <= TOL * abs(loss64)
            agree(part.flat_grad, flat64, "first shard against the fp64 oracle")
    assert abs(l_all - l_sum) <= TOL * abs(l_all)
    agree(g_all, g_sum, "all points against the sum of the shards")
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(5)).cuda()
    l_perm = full(x[perm]).item()
    assert abs(l_perm - l_all) <= TOL * abs(l_all)
    agree(full.flat_grad, g_all, "permuted points")


@pytest.mark.parametrize("sparse", ["1", "0"])
@pytest.mark.parametrize("widths,G,N", [([2, 64, 64, 1], 20, 777), ([2, 32, 256, 64, 1], 20, 1040),
                                        ([2, 64, 256, 64, 1], 10, 515), ([2, 128, 192, 1], 20, 130)])
def test_raw_row_sparse_layers_match_oracle(widths, G, N, sparse, monkeypatch):
    """Hidden layers the window tables do not take (grid 20: 26 intervals; 192 or 256 outputs) run the sparse
    forward / input-gradient kernels on raw weight rows, 64 outputs per block; wider layers take several passes
    over the outputs, each adding to the input gradients.  sparse = "0": the same models with the sparse kernels
    switched off (dense matrix kernels, or streaming where those do not apply)."""
    if sparse == "0":
        monkeypatch.setenv("PDEKAN_DISABLE_SPARSE", "1")
    m = _random_model(widths, G, 300 + N)
    x = _points_clear_of_knots(m, N, "silu", 12)
    _grads_vs_oracle(m, x, "silu", widths)


@pytest.mark.parametrize("widths,G,N", [([6, 64, 64, 1], 10, 1000), ([6, 32, 128, 64, 1], 20, 2064),
                                        ([6, 64, 64, 1], 10, 16)])
def test_basket_weight_gradients_in_five_four_jet_passes(widths, G, N, monkeypatch):
    """6 inputs (5 assets + time): 12 jets per value.  The hidden layers' weight gradients are taken as 5 passes of
    the 4-jet matrix kernels over gathered views (value, first k, time in pass 0, second k), the cotangent's value jet
    in pass 0 only: against the fp64 oracle, and against the streaming 12-jet kernel on the same inputs."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model(widths, G, 500 + N)
    x = _points_clear_of_knots(m, N, "silu", 21)
    step = _grads_vs_oracle(m, x, "silu", widths)
    g_split = step.flat_grad.clone()
    monkeypatch.setenv("PDEKAN_DISABLE_JET_SPLIT", "1")
    step1 = BlackScholesStep(m.cuda())
    step1(x.cuda())
    monkeypatch.delenv("PDEKAN_DISABLE_JET_SPLIT")
    assert max_rel(g_split.cpu(), step1.flat_grad.cpu()) <= 3e-6
    assert not torch.equal(g_split, step1.flat_grad)        # the switch selected something
    if N > 400:                                              # micro-batches: the passes accumulate across them
        mb = BlackScholesStep(m.cuda(), micro_batch=400)
        loss_mb = mb(x.cuda()).item()
        assert abs(loss_mb - step.loss.item()) <= 1e-6 * abs(loss_mb)
        assert max_rel(mb.flat_grad.cpu(), g_split.cpu()) <= 3e-6


def test_split_bf16_and_fp32_blocks_share_one_layer():
    """A 96-input layer with three kinds of knot rows: features 0..31 equally spaced (split-bf16 kernel, polynomial
    pieces), 32..63 with a knot moved by h/100 (split-bf16 kernel, general triangle around the quotient guess),
    64..95 with a knot moved by 0.35 h (fp32 matrix kernel, interval search), all in the same launch group."""
    m = _random_model([2, 96, 64, 1], 10, 90)
    with torch.no_grad():
        m.layers[1].grid[32:64, 7] += 2e-3
        m.layers[1].grid[40, 3] -= 1e-3
        m.layers[1].grid[64:, 7] += 0.07
    x = _points_clear_of_knots(m, 2048, "silu", 91)
    _grads_vs_oracle(m, x, "silu", [2, 96, 64, 1])


@pytest.mark.parametrize("widths,G,N", [([600, 2, 1], 5, 7), ([3000, 2, 1], 5, 1), ([300, 3, 1], 64, 5)])
def test_wide_input_layer_matches_oracle(widths, G, N):
    """Layers wider than one LDS window of the streaming kernels (the branch
    network KAN([3000, 2, 1]) of run1.py:481, evaluated on a single row as all its
    rows are equal, run1.py:112): values, input and parameter gradients."""
    from pde_amd import KAN
    torch.manual_seed(5)
    m = KAN(widths, grid_size=G, base_activation=torch.nn.Identity)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    layers = O.layers_from_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    layers64 = O.layers_from_state_dict({k: v.clone() for k, v in m.state_dict().items()}, dtype=torch.float64)
    m = m.cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(N, widths[0], generator=g) * 2.4 - 1.2          # some inputs fall outside the grid range
    gy = torch.linspace(-1.0, 1.0, N).view(N, 1)
    ref = O.value_step(x.double(), layers64, "identity", gy.double())
    ref32 = O.value_step(x, layers, "identity", gy)
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    (y * gy.cuda()).sum().backward()

    def close(val, r64, r32, what):
        v = val.detach().cpu()
        assert max_rel(v, r64) <= TOL, (what, max_rel(v, r64))
        assert max_rel(v, r32) <= TOL + max_rel(r32, r64), (what, max_rel(v, r32))

    close(y, ref["y"], ref32["y"], "y")
    close(xg.grad, ref["gx"], ref32["gx"], "gx")
    for l, layer in enumerate(m.layers):
        for key in ("base_weight", "spline_weight", "spline_scaler"):
            close(getattr(layer, key).grad, ref["grads"][l][key], ref32["grads"][l][key], f"{l}.{key}")


def test_lbfgs_closure_on_the_fused_step():
    """The LBFGS scripts (run2.py:174-195) drive the model through a closure that
    zeroes the gradients, evaluates the loss and back-propagates.  With the fused
    step as the closure body torch.optim.LBFGS (strong Wolfe, as the scripts
    ask for) must see ordinary .grad tensors and reduce the loss."""
    from pde_amd import KAN
    from pde_amd.pinn import BlackScholesStep
    torch.manual_seed(3)
    m = KAN([2, 8, 1], grid_size=5).cuda()
    x = torch.rand(2048, 2, generator=torch.Generator().manual_seed(4)).cuda()
    step = BlackScholesStep(m)
    opt = torch.optim.LBFGS(m.parameters(), lr=1.0, history_size=10, line_search_fn="strong_wolfe")
    calls = []

    def closure():
        opt.zero_grad()                    # set_to_none: the step re-attaches its gradient views
        loss = step(x)
        calls.append(loss.item())
        return loss

    first = closure().item()
    for _ in range(5):
        opt.step(closure)
    last = step(x).item()
    assert len(calls) > 6 and all(v == v for v in calls)
    assert last < 0.5 * first
    # the gradients the optimiser consumed are the step's flat buffer
    for p in m.parameters():
        assert p.grad is not None and p.grad.data_ptr() >= step.flat_grad.data_ptr()


def test_update_grid_on_device_matches_reference_fixture(golden):
    """`model(x, update_grid=True)` (kan.py:274-279 with :168-215 per layer) on the
    GPU: replays the call that produced the `updated_grid` fixture; both layers'
    grids and refitted spline weights must come out as the reference's."""
    from pde_amd import KAN
    rec = golden("updated_grid")
    seed = int(rec["seed"])
    torch.manual_seed(seed)
    m = KAN([2, 8, 1], grid_size=5)
    gen = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.1 * torch.randn(p.shape, generator=gen))
    torch.rand(128, 2, generator=gen)                    # the fixture's points were drawn first
    torch.rand(12 + 6, 2, generator=gen)
    samples = torch.rand(256, 2, generator=gen) * 1.4 - 0.2
    m = m.cuda()
    with torch.no_grad():
        m(samples.cuda(), update_grid=True)
    sd = golden_state_dict(rec)
    for l in range(2):
        assert max_rel(m.layers[l].grid.cpu(), sd[f"layers.{l}.grid"]) <= 1e-5, l
        # (the refit is solved in fp64 by the library's own kernel: the design matrices of this fixture have condition
        # numbers 36 .. 71, the reference's fp32 lstsq sits 2e-6 / 9e-6 from the fp64 solution of the same system)
        assert max_rel(m.layers[l].spline_weight.cpu(), sd[f"layers.{l}.spline_weight"]) <= 5e-5, l
    # and the step on the moved grid still matches the fixture's numbers
    from pde_amd.pinn import BlackScholesStep
    m.load_state_dict(sd)
    x = torch.from_numpy(rec["x"]).cuda()
    loss = BlackScholesStep(m)(x)
    check(loss, rec, "loss")


@pytest.mark.parametrize("n,in_f,out_f,G", [(256, 2, 8, 5), (1000, 8, 1, 5), (700, 16, 300, 10), (6, 3, 5, 5),
                                             (11, 4, 7, 10), (513, 5, 64, 20)])
def test_curve2coeff_on_device_is_the_fp64_least_squares_solution(n, in_f, out_f, G):
    """`curve2coeff` (kan.py:113-143) through pdekan_fit_coefficients: against the minimal-norm least-squares solution
    of the same system in fp64 (numpy lstsq on the dense bases of the CPU code), for well-determined systems,
    for fewer samples than coefficients (the initialisation's case, kan.py:67-73: G+1 samples, G+3 unknowns), for
    samples that leave columns of the design matrix empty, with y a permuted view, and more than 256 outputs."""
    from pde_amd import KANLinear
    from pde_amd.spline_host import dense_bsplines
    gen = torch.Generator().manual_seed(n * 31 + G)
    lay = KANLinear(in_f, out_f, grid_size=G)
    x = torch.rand(n, in_f, generator=gen) * 1.6 - 0.8 if n > 20 else torch.rand(n, in_f, generator=gen) * 0.4
    y_t = torch.randn(in_f, n, out_f, generator=gen)          # stored (in, batch, out): y is a permuted view
    y = y_t.permute(1, 0, 2)
    lay = lay.cuda()
    got = lay.curve2coeff(x.cuda(), y_t.cuda().permute(1, 0, 2)).cpu()
    A = dense_bsplines(x, lay.grid.cpu(), 3).transpose(0, 1).double().numpy()      # [in, batch, C]
    B = y.transpose(0, 1).double().numpy()
    rcond = float(np.finfo(np.float32).eps) * max(n, G + 3)
    want = np.stack([np.linalg.lstsq(A[i], B[i], rcond=rcond)[0] for i in range(in_f)], 0)   # [in, C, out]
    want = torch.from_numpy(want).permute(2, 0, 1)
    assert got.shape == want.shape
    err = ((got.double() - want).abs().max() / want.abs().max()).item()
    # the fitted curve at the samples
    fit = np.einsum("ipc,oic->ipo", A, got.double().numpy())
    ref = np.einsum("ipc,oic->ipo", A, want.numpy())
    cerr = np.abs(fit - ref).max() / max(np.abs(ref).max(), 1e-30)
    # (fewer samples than coefficients: the smallest retained singular value is small, and the 1e-7 between the
    # kernel's basis values and the CPU code's is amplified by it)
    tol = 2e-5 if n > G + 3 else 5e-4
    assert err <= tol and cerr <= tol, (err, cerr)


def test_jacrev_hessian_callers_work_on_the_gpu(golden):
    """run1.py:121-124 as written (jacrev / hessian through `forward`) on a GPU model:
    same numbers as the fixture and as the O(n) jets."""
    from torch.func import hessian, jacrev
    rec = golden("jacrev_hessian")
    m = build_model(rec)
    x = torch.from_numpy(rec["x"][:, 0].copy()).cuda()
    t = torch.from_numpy(rec["x"][:, 1].copy()).cuda()

    def net(xx, tt):
        return m(torch.stack([xx.reshape(-1), tt.reshape(-1)]).permute(1, 0))[:, 0]

    s = net(x, t)                                        # the kernels
    s_x = jacrev(net, argnums=0)(x, t).sum(dim=0)        # tensor ops under the transform
    s_xx = hessian(net, argnums=0)(x, t).sum(dim=0).sum(dim=0)
    s_t = jacrev(net, argnums=1)(x, t).sum(dim=0)
    for val, key in ((s, "s"), (s_x, "s_x"), (s_xx, "s_xx"), (s_t, "s_t")):
        assert max_rel(val.detach().cpu(), rec[key]) <= TOL, key
    jet = m.forward_jet(torch.stack([x, t], 1))[:, 0, :]
    assert max_rel(jet[:, 1].cpu(), s_x.cpu()) <= TOL
    assert max_rel(jet[:, 2].cpu(), s_t.cpu()) <= TOL
    assert max_rel(jet[:, 3].cpu(), s_xx.cpu()) <= TOL


@pytest.mark.parametrize("widths,G", [([2, 64, 64, 1], 10), ([2, 64, 128, 64, 1], 10), ([2, 64, 64, 1], 20)])
def test_fused_step_is_bitwise_reproducible(widths, G):
    """No float atomics anywhere and fixed-order reductions: identical inputs give
    bit-identical loss and gradients, at a size where every CU holds several
    workgroups of every kernel."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model(widths, G, 77).cuda()
    x = torch.rand(1 << 18, 2, generator=torch.Generator().manual_seed(9)).cuda()
    step = BlackScholesStep(m)
    l0 = step(x).clone()
    g0 = step.flat_grad.clone()
    for _ in range(3):
        l1 = step(x)
        assert torch.equal(l1, l0)
        assert torch.equal(step.flat_grad, g0)


def test_fused_step_replays_from_a_hip_graph():
    """The whole step (every kernel path of a [2,64,64,1] grid-10 model, the device-side choice between the
    fp32 and the split-bf16 weight-gradient kernels included) captured once and replayed on new points: same
    bits as the eager call.  Nothing in the library allocates, synchronises or reads back during a step."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model([2, 64, 64, 1], 10, 5).cuda()
    step = BlackScholesStep(m)
    gen = torch.Generator().manual_seed(3)
    xs = [torch.rand(4096, 2, generator=gen).cuda() for _ in range(3)]
    x_static = xs[0].clone()
    step(x_static)                                   # warm-up: workspace allocation happens here
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(x_static)
    for x in xs:
        x_static.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        loss_g, grad_g = step.loss.item(), step.flat_grad.clone()
        step(x)
        assert step.loss.item() == loss_g
        assert torch.equal(step.flat_grad, grad_g)


@pytest.mark.parametrize("I,O,N,G,knots", [
    (32, 64, 48, 10, "default"), (96, 64, 1000, 10, "default"), (128, 128, 4112, 10, "default"),
    (256, 256, 1000, 10, "default"), (64, 256, 2064, 10, "default"), (32, 128, 16, 10, "default"),
    (32, 64, 48, 20, "exact"), (48, 64, 1000, 20, "exact"), (256, 256, 2064, 20, "default"),
    (64, 128, 4099, 20, "default"), (32, 64, 64, 20, "default"), (64, 64, 1000, 10, "moved")])
def test_split_bf16_weight_gradients_over_layer_shapes(I, O, N, G, knots, monkeypatch):
    """One grid-10 or grid-20 layer at a time through the layer-level entry points (jet forward + backward with a random
    cotangent): every block / chunk decode of the split-bf16 weight-gradient kernel (1 to 8 feature blocks, 1 to 4
    output tiles, ragged N with the fp32 tail chunk) against the fp32 matrix kernel on the same inputs.  Knot rows:
    the default grid 10 is equally spaced (polynomial pieces), the default grid 20 rounds its knots 1.5e-6 h off
    (the kernel's general-knot variant), "exact" = grid 20 on [-1.25, 1.25], h = 1/8 (polynomial pieces again),
    "moved" = a grid-10 row with two knots moved by up to h/20 (general-knot variant)."""
    from pde_amd.kan import KANLinear
    torch.manual_seed(I + O)
    lay = KANLinear(I, O, grid_size=G, grid_range=[-1.25, 1.25] if knots == "exact" else [-1, 1]).cuda()
    if knots == "moved":
        with torch.no_grad():
            lay.grid[:, 5] += 0.01
            lay.grid[::3, 9] -= 0.004
    g = torch.Generator().manual_seed(N)
    xj = torch.randn(N, I, 4, generator=g)
    xj[:, :, 0] = torch.rand(N, I, generator=g) * 3.6 - 1.8          # also outside the extended grid
    gy = torch.randn(N, O, 4, generator=g).cuda()
    xj = xj.cuda()
    grads = []
    for off in (None, "1"):
        if off:
            monkeypatch.setenv("PDEKAN_DISABLE_BF16X3", off)
        for p in lay.parameters():
            p.grad = None
        lay.forward_jet(xj, 2, 1).backward(gy)
        grads.append([p.grad.clone() for p in lay.parameters()])
    monkeypatch.delenv("PDEKAN_DISABLE_BF16X3")
    # heavy-tailed random jets: each kernel's fp32 accumulation alone sits 1e-6 to 3e-6 from an fp64 sum here
    for a, b in zip(*grads):
        assert max_rel(a.cpu(), b.cpu()) <= 5e-6
    # the two kernels sum in different orders: bit-identical results would mean the switch selected nothing
    assert not torch.equal(grads[0][1], grads[1][1])
    x = xj.double().cpu()
    x0, sg = x[..., 0], torch.sigmoid(x[..., 0])
    f0, f1, f2 = x0 * sg, sg * (1 + x0 * (1 - sg)), sg * (1 - sg) * (2 + x0 * (1 - 2 * sg))
    aj = torch.stack([f0, f1 * x[..., 1], f1 * x[..., 2], f2 * x[..., 1] ** 2 + f1 * x[..., 3]], -1)
    exact = torch.einsum("noj,nij->oi", gy.double().cpu(), aj)
    for gset in grads:
        assert max_rel(gset[0].double().cpu(), exact) <= 5e-6
