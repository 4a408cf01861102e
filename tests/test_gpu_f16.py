"""The fp16 two-piece matrix-core kernels (forward, input gradients, weight gradients of hidden layers with
(nearly) equally spaced knots) against the fp32 kernels on the same inputs and against fp64 tensor math, one layer
at a time through the layer-level entry points of the C ABI."""
import pytest
import torch

from conftest import max_rel
from oracle import kan_oracle as O

pytestmark = pytest.mark.gpu


def _layer(I, O_, G, knots, seed):
    from pde_amd.kan import KANLinear
    torch.manual_seed(seed)
    lay = KANLinear(I, O_, grid_size=G, grid_range=[-1.25, 1.25] if knots == "exact" else [-1, 1]).cuda()
    with torch.no_grad():
        for p in lay.parameters():
            p.add_(0.1 * torch.randn_like(p))
        if knots == "jittered":                       # still within h/4 of equal spacing: the nearly-equal variant
            lay.grid[:, 5] += 0.01
            lay.grid[::3, 9] -= 0.004
        if knots == "moved":                          # far from equal spacing: the kernels must step aside
            lay.grid[:, 6] += 0.09
    return lay


def _run(lay, xj, gy):
    for p in lay.parameters():
        p.grad = None
    x = xj.clone().requires_grad_(True)
    y = lay.forward_jet(x, 2, 1)
    y.backward(gy)
    return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in lay.parameters()]


def _jets(N, I, O_, seed, jet_scale=(1.0, 1.0, 1.0, 1.0), cot_scale=(1.0, 1.0, 1.0, 1.0)):
    g = torch.Generator().manual_seed(seed)
    xj = torch.randn(N, I, 4, generator=g) * torch.tensor(jet_scale)
    xj[:, :, 0] = torch.rand(N, I, generator=g) * 3.6 - 1.8          # also outside the extended grid
    gy = torch.randn(N, O_, 4, generator=g) * torch.tensor(cot_scale)
    return xj.cuda(), gy.cuda()


NAMES = ["y", "gx", "base_weight", "spline_weight", "spline_scaler"]


@pytest.mark.parametrize("I,O_,N,G,knots", [
    (64, 64, 4099, 10, "default"), (8, 64, 48, 10, "default"), (16, 128, 1000, 10, "default"),
    (128, 128, 2064, 10, "default"), (256, 256, 1024, 10, "default"), (64, 256, 333, 10, "default"),
    (32, 64, 17, 10, "default"), (64, 64, 15, 10, "default"), (64, 64, 1000, 10, "jittered"),
    (64, 64, 2000, 20, "default"), (32, 64, 48, 20, "exact"), (256, 256, 1040, 20, "default"),
    (64, 128, 4099, 20, "default"), (48, 192, 500, 20, "exact")])
def test_fp16_kernels_agree_with_the_fp32_kernels(I, O_, N, G, knots, monkeypatch):
    """Every block / pass decode of the three kernels (1 to 4 output blocks, 2 to 64 stages, ragged N with the
    sparse-kernel tail, the accumulator flush of wide layers) against the fp32 vector / matrix kernels.  Knot rows:
    default grid 10 = equally spaced; default grid 20 and "jittered" = nearly (general triangle); "exact" = grid 20
    on [-1.25, 1.25], equally spaced."""
    lay = _layer(I, O_, G, knots, I + O_)
    xj, gy = _jets(N, I, O_, N)
    a = _run(lay, xj, gy)
    monkeypatch.setenv("PDEKAN_DISABLE_F16", "1")
    b = _run(lay, xj, gy)
    monkeypatch.delenv("PDEKAN_DISABLE_F16")
    for name, u, v in zip(NAMES, a, b):
        # heavy-tailed random jets: each kernel's fp32 accumulation alone sits 1e-6 to 3e-6 from an fp64 sum here
        assert max_rel(u.cpu(), v.cpu()) <= 5e-6, (name, max_rel(u.cpu(), v.cpu()))
    assert not torch.equal(a[0], b[0])                                      # the switch selected something
    if N >= 16:                                                             # (fewer points: all of them are the sparse-kernel tail)
        assert not torch.equal(a[1], b[1])
    if I >= 32 and N >= 1000:                                            # (smaller N: one chunk, fp32 matrix kernel either way)
        assert not torch.equal(a[3], b[3])


def test_knot_rows_far_from_equal_spacing_fall_back_bit_for_bit(monkeypatch):
    lay = _layer(64, 64, 10, "moved", 3)
    xj, gy = _jets(1000, 64, 64, 4)
    a = _run(lay, xj, gy)
    monkeypatch.setenv("PDEKAN_DISABLE_F16", "1")
    b = _run(lay, xj, gy)
    monkeypatch.delenv("PDEKAN_DISABLE_F16")
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("jet_scale,cot_scale", [
    ((1.0, 1e3, 1e-3, 1e6), (1e-9, 1e-6, 1e-12, 1e-9)),       # second derivatives 10^6 above values, tiny cotangents
    ((1.0, 1e-4, 1e4, 1e-8), (1e5, 1e-3, 1e2, 1e6)),
    ((1.0, 1.0, 1.0, 1.0), (1.0, 0.0, 0.0, 0.0)),              # whole jet slots of the cotangent zero
])
def test_per_jet_scales_cover_wide_dynamic_ranges(jet_scale, cot_scale):
    """The operands are scaled per jet slot into fp16's range before they are cut in two; jets that differ by many
    orders of magnitude (and slots that are zero throughout) must neither overflow nor lose their small slots.
    Against fp64 tensor math on the same layer, slot by slot."""
    I, O_, N, G = 64, 64, 1024, 10
    lay = _layer(I, O_, G, "default", 9)
    xj, gy = _jets(N, I, O_, 10, jet_scale, cot_scale)
    y, gx, gb, gs, gc = _run(lay, xj, gy)
    assert all(torch.isfinite(t).all() for t in (y, gx, gb, gs, gc))
    # fp64 reference of the jet forward by autograd on the oracle's dense layer
    p64 = {"grid": lay.grid.double().cpu(), "base_weight": lay.base_weight.detach().double().cpu(),
           "spline_weight": lay.spline_weight.detach().double().cpu(),
           "spline_scaler": lay.spline_scaler.detach().double().cpu()}
    leaves = {k: p64[k].clone().requires_grad_(True) for k in ("base_weight", "spline_weight", "spline_scaler")}
    q = dict(p64, **leaves)
    x64 = xj.double().cpu()
    a0 = x64[:, :, 0].clone().requires_grad_(True)
    a1, a2, a3 = (x64[:, :, j].clone().requires_grad_(True) for j in (1, 2, 3))
    # y0 = f(a0); y1 = J a1; y2 = J a2; y3 = a1^T H a1 (diagonal: per-feature second derivative) + J a3, with f separable
    f = O.layer_forward(a0, q, "silu")                                   # [N, O]
    yref = torch.zeros(N, O_, 4, dtype=torch.float64)
    cols = []
    for o in range(O_):
        (g1,) = torch.autograd.grad(f[:, o].sum(), a0, create_graph=True)           # d f_o / d a0_i  [N, I]
        (g2,) = torch.autograd.grad((g1 * a1).sum(), a0, create_graph=True)      # f''_oi a1_i per feature (f is separable)
        cols.append(torch.stack([f[:, o], (g1 * a1).sum(1), (g1 * a2).sum(1), (g2 * a1).sum(1) + (g1 * a3).sum(1)], -1))
    yfull = torch.stack(cols, 1)                                          # [N, O, 4], differentiable
    for j in range(4):
        assert max_rel(y[:, :, j].cpu(), yfull[:, :, j].detach()) <= 1e-5, ("y", j)
    loss = (yfull * gy.double().cpu()).sum()
    grads = torch.autograd.grad(loss, [a0, a1, a2, a3] + [leaves[k] for k in ("base_weight", "spline_weight", "spline_scaler")])
    gxref = torch.stack(grads[:4], -1)
    for j in range(4):
        if gxref[:, :, j].abs().max() > 0:
            assert max_rel(gx[:, :, j].cpu(), gxref[:, :, j]) <= 1e-5, ("gx", j)
    for name, got, ref in zip(NAMES[2:], (gb, gs, gc), grads[4:]):
        assert max_rel(got.cpu(), ref) <= 1e-5, name


# ---- 12 jets (5 assets + time) ----------------------------------------------------------------------------
def _run12(lay, xj, gy):
    for p in lay.parameters():
        p.grad = None
    x = xj.clone().requires_grad_(True)
    y = lay.forward_jet(x, 6, 5)
    y.backward(gy)
    return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in lay.parameters()]


@pytest.mark.parametrize("I,O_,N,knots", [(64, 64, 70000, "default"), (8, 64, 5, "default"), (256, 256, 1030, "default"),
                                          (128, 192, 4099, "jittered"), (24, 64, 640001, "default")])
def test_forward_on_the_sparse_matrix_instruction_agrees_with_the_dense_one(I, O_, N, knots, monkeypatch):
    """The grid-10 forward takes the 5 live columns of a feature as a 2:4 structured-sparse operand
    (v_smfmac_f32_16x16x64_f16, operands straight from the expansion's registers, its own image of the W planes);
    with PDEKAN_DISABLE_F16_SPARSE=1 the dense 16 x 16 x 32 kernel and its image run.  The products are the same,
    the order of the sums is not.  Points outside the extended grid and in its first / last intervals (live columns
    outside the layer's columns), 2 to 64 stages, the accumulator flush of wide layers, more tiles than blocks
    (640001 points: blocks walk several tiles, the last one ragged); the backward kernels are the same either way."""
    lay = _layer(I, O_, 10, knots, I + O_ + 7)
    xj, gy = _jets(N, I, O_, N + 3)
    a = _run(lay, xj, gy)
    monkeypatch.setenv("PDEKAN_DISABLE_F16_SPARSE", "1")
    b = _run(lay, xj, gy)
    monkeypatch.delenv("PDEKAN_DISABLE_F16_SPARSE")
    for j in range(4):
        assert max_rel(a[0][..., j].cpu(), b[0][..., j].cpu()) <= 2e-6, ("y", j)
    assert not torch.equal(a[0], b[0])
    for u, v in zip(a[1:], b[1:]):
        assert torch.equal(u, v)


def _jets12(N, I, O_, seed, scale=None):
    g = torch.Generator().manual_seed(seed)
    s = torch.ones(12) if scale is None else torch.tensor(scale)
    xj = torch.randn(N, I, 12, generator=g) * s
    xj[:, :, 0] = torch.rand(N, I, generator=g) * 3.6 - 1.8          # also outside the extended grid
    gy = torch.randn(N, O_, 12, generator=g)
    return xj.cuda(), gy.cuda()


@pytest.mark.parametrize("I,O_,N,knots", [(64, 64, 1000, "default"), (128, 128, 2064, "default"), (32, 64, 16, "default"),
                                          (32, 256, 535, "default"), (256, 64, 48, "default"), (64, 128, 4100, "jittered"),
                                          (64, 64, 7, "default")])
def test_fp16_kernels_with_12_jets_agree_with_the_streaming_kernels(I, O_, N, knots, monkeypatch):
    """Layer-level calls with 6 directions and 5 second derivatives: 8 points x 3 slot groups per wave (forward,
    input gradients), virtual points (weight gradients), 1 to 4 output blocks, the accumulator flush of wide layers,
    ragged N (the last N % 8 resp. N % 16 points stream), against the streaming 12-jet kernels."""
    lay = _layer(I, O_, 10, knots, I + O_ + 1)
    xj, gy = _jets12(N, I, O_, N)
    a = _run12(lay, xj, gy)
    monkeypatch.setenv("PDEKAN_DISABLE_F16", "1")
    b = _run12(lay, xj, gy)
    monkeypatch.delenv("PDEKAN_DISABLE_F16")
    for name, u, v in zip(NAMES, a, b):
        assert max_rel(u.cpu(), v.cpu()) <= 5e-6, (name, max_rel(u.cpu(), v.cpu()))
    if N >= 16:
        assert not torch.equal(a[0], b[0]) and not torch.equal(a[1], b[1]) and not torch.equal(a[3], b[3])


@pytest.mark.parametrize("I,O_,N,knots", [(128, 128, 2064, "default"), (64, 64, 8, "default"), (96, 256, 535, "jittered")])
def test_12_jet_forward_on_wide_tiles_agrees_with_the_narrow_tiles(I, O_, N, knots, monkeypatch):
    """The 12-jet forward runs on 32 x 32 x 16 matrix tiles with its own image of the W planes; with
    PDEKAN_DISABLE_F16_WIDE=1 the 16 x 16 x 32 kernel and its image are used.  Same outputs to fp32 reassociation
    (the products are the same, the order of the sums is not); the backward kernels are the same either way."""
    lay = _layer(I, O_, 10, knots, I + O_ + 3)
    xj, gy = _jets12(N, I, O_, N + 2)
    monkeypatch.setenv("PDEKAN_DISABLE_F16_SPARSE", "1")      # (layers this narrow would take the sparse-operand kernel)
    a = _run12(lay, xj, gy)
    monkeypatch.setenv("PDEKAN_DISABLE_F16_WIDE", "1")
    b = _run12(lay, xj, gy)
    monkeypatch.delenv("PDEKAN_DISABLE_F16_WIDE")
    monkeypatch.delenv("PDEKAN_DISABLE_F16_SPARSE")
    for j in range(12):
        assert max_rel(a[0][..., j].cpu(), b[0][..., j].cpu()) <= 2e-6, ("y", j)
    assert not torch.equal(a[0], b[0])
    for u, v in zip(a[1:], b[1:]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("I,O_,N,knots", [(128, 128, 70000, "default"), (32, 64, 8, "default"), (160, 256, 1031, "jittered"),
                                          (64, 64, 140008, "default")])
def test_12_jet_forward_on_the_sparse_matrix_instruction_agrees_with_the_dense_one(I, O_, N, knots, monkeypatch):
    """The 12-jet forward of layers up to 160 inputs takes the live columns as a 2:4 structured-sparse operand (a row
    tile = one slot of 16 points, operands rewritten in place from the expansion's registers); with
    PDEKAN_DISABLE_F16_SPARSE=1 the dense 32 x 32 x 16 kernel and its image run.  Same products, another order of the
    sums.  Points outside the extended grid, 8 to 40 stages, ragged last tiles, more tiles than blocks (140008
    points: blocks walk several tiles); the backward kernels are the same either way."""
    lay = _layer(I, O_, 10, knots, I + O_ + 5)
    xj, gy = _jets12(N, I, O_, N + 4)
    a = _run12(lay, xj, gy)
    monkeypatch.setenv("PDEKAN_DISABLE_F16_SPARSE", "1")
    b = _run12(lay, xj, gy)
    monkeypatch.delenv("PDEKAN_DISABLE_F16_SPARSE")
    for j in range(12):
        assert max_rel(a[0][..., j].cpu(), b[0][..., j].cpu()) <= 2e-6, ("y", j)
    assert not torch.equal(a[0], b[0])
    for u, v in zip(a[1:], b[1:]):
        assert torch.equal(u, v)


def test_12_jet_layers_with_moved_knots_fall_back(monkeypatch):
    """Knot rows far from equal spacing: the matrix kernels step aside on the device and the streaming kernels
    launched beside them do the layer.  Outputs and input gradients are theirs bit for bit; the weight gradients are
    summed over other chunks of points (the last N % 16 points are a chunk of their own), so they agree to rounding."""
    lay = _layer(64, 64, 10, "moved", 5)
    xj, gy = _jets12(1000, 64, 64, 6)
    a = _run12(lay, xj, gy)
    monkeypatch.setenv("PDEKAN_DISABLE_F16", "1")
    b = _run12(lay, xj, gy)
    monkeypatch.delenv("PDEKAN_DISABLE_F16")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for name, u, v in zip(NAMES[2:], a[2:], b[2:]):
        assert max_rel(u.cpu(), v.cpu()) <= 2e-6, name


def test_12_jet_slot_scales_cover_wide_dynamic_ranges(monkeypatch):
    """Slots that differ by many orders of magnitude: every slot of the outputs and gradients stays within the fp32
    bar of the streaming kernels' result, measured slot by slot (a shared scale would wipe out the small slots)."""
    scale = [1.0, 1e3, 1e-3, 1.0, 1e2, 1e-2, 1e-4, 1e6, 1e-6, 1.0, 1e3, 1e-3]
    lay = _layer(64, 64, 10, "default", 11)
    xj, gy = _jets12(1024, 64, 64, 12, scale)
    a = _run12(lay, xj, gy)
    monkeypatch.setenv("PDEKAN_DISABLE_F16", "1")
    b = _run12(lay, xj, gy)
    monkeypatch.delenv("PDEKAN_DISABLE_F16")
    for j in range(12):
        assert max_rel(a[0][..., j].cpu(), b[0][..., j].cpu()) <= 5e-6, ("y", j)
        assert max_rel(a[1][..., j].cpu(), b[1][..., j].cpu()) <= 5e-6, ("gx", j)
    for name, u, v in zip(NAMES[2:], a[2:], b[2:]):
        assert max_rel(u.cpu(), v.cpu()) <= 5e-6, name


@pytest.mark.parametrize("I,O_,G,N,knots", [(64, 64, 10, 70000, "default"), (128, 128, 10, 5000, "default"),
                                            (64, 128, 10, 4099, "jittered"), (192, 64, 10, 2049, "default"),
                                            (64, 64, 20, 70000, "default"), (32, 128, 20, 9000, "jittered"),
                                            (96, 64, 20, 2049, "default"), (256, 256, 20, 4640, "default")])
def test_weight_gradients_on_wide_tiles_agree_with_the_narrow_tiles(I, O_, G, N, knots, monkeypatch):
    """Grid-10 layers whose inputs come in multiples of 64 (grid-20 layers: of 32) take their weight gradients on
    32 x 32 x 16 matrix tiles (16 features x 64 outputs per wave and 4-point steps, or 8 features and 8-point steps; no
    second register set: every 512 points the accumulators go to the chunk's partial in memory, by atomic adds after
    the first time); with PDEKAN_DISABLE_F16_WIDE=1 the 16 x 16 x 32 kernel runs (8 or 4 features per wave).  Same
    gradients to fp32 reassociation, every other output bit for bit (only the weight gradients change kernel); many
    chunks, chunks of several flushes, a ragged last chunk, several feature and output blocks, nearly equally spaced
    knot rows."""
    lay = _layer(I, O_, G, knots, I + O_ + 1)
    xj, gy = _jets(N, I, O_, N + 1)
    a = _run(lay, xj, gy)
    monkeypatch.setenv("PDEKAN_DISABLE_F16_WIDE", "1")
    b = _run(lay, xj, gy)
    monkeypatch.delenv("PDEKAN_DISABLE_F16_WIDE")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for name, u, v in zip(NAMES[2:], a[2:], b[2:]):
        assert max_rel(u.cpu(), v.cpu()) <= 3e-6, (name, max_rel(u.cpu(), v.cpu()))
    assert not torch.equal(a[3], b[3])
    a2 = _run(lay, xj, gy)                                 # and bit for bit the same from run to run
    for u, v in zip(a, a2):
        assert torch.equal(u, v)
