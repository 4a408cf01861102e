"""Edges of the fused step and of the layer entry points on the GPU: workspace bounds, heads too wide for the
fused kernel, layers without a standalone scaler, points that sit on knots, and the data-parallel wrapper on a
one-rank RCCL group.  Everything goes through the C ABI."""
import ctypes
import os

import pytest
import torch

from conftest import max_rel
from oracle import kan_oracle as O
from test_gpu_parity import TOL, _random_model

pytestmark = pytest.mark.gpu
KEYS = ("base_weight", "spline_weight", "spline_scaler")


def _flat(grads):
    return torch.cat([g[k].flatten() for g in grads for k in KEYS if g[k] is not None])


@pytest.mark.parametrize("widths,G,N,mb", [([2, 64, 64, 1], 10, 1 << 20, 0), ([2, 64, 64, 1], 10, 70000, 0),
                                           ([2, 8, 1], 5, 300, 0), ([6, 128, 128, 1], 10, 4107, 0),
                                           ([6, 128, 128, 1], 10, 4107, 1000), ([2, 64, 64, 1], 10, 70001, 4096)])
def test_step_stays_inside_the_workspace_it_asked_for(widths, G, N, mb):
    """The step is given exactly the bytes pdekan_model_workspace_bytes() names, inside a larger allocation
    whose remainder holds a pattern: every kernel of the step (the fused head writes one loss partial per
    block, up to 1280 blocks at this size; the 12-jet layers lay out matrix chunks, a tail chunk, the chunks of
    whatever runs instead of the matrix kernels, three 4-jet views and 12-word maxima) must leave that remainder
    alone, and the result must equal the ordinary call's.  N % 16 != 0 and micro-batches exercise the tails."""
    from pde_amd import _lib, functional as K
    from pde_amd.pinn import BlackScholesStep
    lib = _lib.load()
    d = widths[0]
    m = _random_model(widths, G, 5).cuda()
    x = torch.rand(N, d, generator=torch.Generator().manual_seed(6)).cuda()
    step = BlackScholesStep(m, micro_batch=mb)
    step(x)
    loss0, grad0 = step.loss.clone(), step.flat_grad.clone()
    layers, grads, nl = step._structs()
    need = lib.pdekan_model_workspace_bytes(layers, nl, N, d, d - 1, mb)
    guard = 1 << 16
    buf = torch.empty(need + guard, dtype=torch.uint8, device="cuda")
    buf[need:] = 0xA5
    step.flat_grad.zero_()
    _lib.check(lib.pdekan_bs_step(layers, nl, K._ptr(x), x.stride(0), x.stride(1), None, N, N, step.sigma, step.rate,
                                  K._ptr(step.loss), None, grads, 0, mb, K._ptr(buf), need, K._stream(x.device)))
    torch.cuda.synchronize()
    assert bool((buf[need:] == 0xA5).all()), "a kernel wrote past the end of the workspace"
    assert torch.equal(step.loss, loss0) and torch.equal(step.flat_grad, grad0)
    # one byte less is refused, not overrun
    rc = lib.pdekan_bs_step(layers, nl, K._ptr(x), x.stride(0), x.stride(1), None, N, N, step.sigma, step.rate,
                            K._ptr(step.loss), None, grads, 0, mb, K._ptr(buf), need - 257, K._stream(x.device))
    assert rc == -2


@pytest.mark.parametrize("moved", [0.0, 0.015, 0.08])
@pytest.mark.parametrize("I,O_,N", [(128, 128, 1003), (32, 64, 40)])
def test_layer_backward_with_12_jets_stays_inside_its_workspace(I, O_, N, moved):
    """pdekan_layer_backward with (n_dir, n_second) = (6, 5), given exactly pdekan_layer_workspace_bytes(): matrix
    chunks, the tail chunk and the streaming kernels' chunks share the partial-sum area; equally spaced knot rows
    (moved = 0) or nearly so (0.015 of a spacing of 0.2) take the 12-jet matrix kernels, rows moved further (0.08) the
    streaming kernels launched beside them.  The bytes behind the
    workspace must survive and both runs must agree with a call on an oversized workspace."""
    from pde_amd import _lib, functional as K
    from pde_amd._lib import PdekanLayerGrad
    from pde_amd.kan import KANLinear
    lib = _lib.load()
    torch.manual_seed(I + O_)
    lay = KANLinear(I, O_, grid_size=10).cuda()
    if moved:
        with torch.no_grad():
            lay.grid[:, 3:-3] += 2 * moved * torch.rand(I, 11, device="cuda") - moved
    gen = torch.Generator().manual_seed(N)
    xj = (torch.rand(N, I, 12, generator=gen) * 2 - 1).cuda()
    gy = torch.randn(N, O_, 12, generator=gen).cuda()
    L = K.layer_struct(lay.grid, lay.base_weight.detach(), lay.spline_weight.detach(), lay.spline_scaler.detach(), 1, 10, 3)
    need = lib.pdekan_layer_workspace_bytes(ctypes.byref(L), N, 6, 5)
    guard = 1 << 16

    def run(nbytes_given, total):
        buf = torch.empty(total, dtype=torch.uint8, device="cuda")
        buf[nbytes_given:] = 0xA5
        gx = torch.empty_like(xj)
        g = [torch.empty_like(lay.base_weight), torch.empty_like(lay.spline_weight), torch.empty_like(lay.spline_scaler)]
        G = PdekanLayerGrad(g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr())
        _lib.check(lib.pdekan_layer_backward(ctypes.byref(L), K._ptr(xj), K._ptr(gy), N, 6, 5, K._ptr(gx),
                                             ctypes.byref(G), 0, K._ptr(buf), nbytes_given, K._stream(xj.device)))
        torch.cuda.synchronize()
        assert bool((buf[nbytes_given:] == 0xA5).all()), "a kernel wrote past the end of the workspace"
        return [gx] + g

    exact = run(need, need + guard)
    roomy = run(4 * need, 4 * need + guard)
    for a, b in zip(exact, roomy):
        assert torch.equal(a, b)


@pytest.mark.parametrize("widths,G", [([2, 256, 1], 64), ([2, 256, 1], 48), ([2, 200, 1], 40)])
def test_heads_too_large_for_the_fused_kernel_take_the_separate_kernels(widths, G):
    """256 inputs at grid 48 / 64 need more LDS than a workgroup has (163,844 B / 213 KB): the step must fall back
    to last-layer forward + residual + backward kernels instead of failing the launch."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model(widths, G, 11)
    x = torch.rand(257, 2, generator=torch.Generator().manual_seed(12))
    ref = O.pinn_step(x.double(), O.layers_from_state_dict(m.state_dict(), dtype=torch.float64), "silu")
    mg = m.cuda()
    step = BlackScholesStep(mg)
    res = torch.empty(257, device="cuda")
    loss = step(x.cuda(), residual_out=res).item()
    assert max_rel(res.cpu(), ref["res"]) <= TOL
    assert abs(loss - ref["loss"].item()) <= TOL * abs(ref["loss"].item())
    assert max_rel(step.flat_grad.cpu(), _flat(ref["grads"])) <= TOL


@pytest.mark.parametrize("widths,G,N,act", [([2, 256, 1], 20, 1000, "silu"), ([2, 192, 1], 30, 37, "silu"),
                                            ([2, 256, 1], 20, 5, "identity"), ([2, 64, 256, 1], 20, 4099, "silu")])
def test_wide_heads_whose_waves_share_a_gradient_table(widths, G, N, act, monkeypatch):
    """Heads whose gradient table is too large for one per wave (256 inputs at grid 20, 192 at grid 30): the waves
    of a group take 64 features each of the same point and share a table.  Against the fp64 oracle and against
    the table-per-wave kernel; ragged point counts leave groups idle in the last iteration."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model(widths, G, 21, act=torch.nn.SiLU if act == "silu" else torch.nn.Identity)
    x = torch.rand(N, 2, generator=torch.Generator().manual_seed(22))
    tgt = torch.rand(N, generator=torch.Generator().manual_seed(23))
    ref = O.pinn_step(x.double(), O.layers_from_state_dict(m.state_dict(), dtype=torch.float64), act, target=tgt.double())
    mg = m.cuda()
    step = BlackScholesStep(mg)
    res = torch.empty(N, device="cuda")
    loss = step(x.cuda(), target=tgt.cuda(), residual_out=res).item()
    assert max_rel(res.cpu(), ref["res"]) <= TOL
    assert abs(loss - ref["loss"].item()) <= TOL * abs(ref["loss"].item())
    assert max_rel(step.flat_grad.cpu(), _flat(ref["grads"])) <= TOL
    g_split, l_split = step.flat_grad.clone(), step.loss.clone()
    step(x.cuda(), target=tgt.cuda())
    assert torch.equal(step.flat_grad, g_split) and torch.equal(step.loss, l_split)      # same bits every time
    monkeypatch.setenv("PDEKAN_DISABLE_HEAD_SPLIT", "1")
    step2 = BlackScholesStep(mg)
    step2(x.cuda(), target=tgt.cuda())
    assert max_rel(step2.flat_grad.cpu(), g_split.cpu()) <= 2e-6
    assert not torch.equal(step2.flat_grad, g_split)                # (another summation order: another kernel ran)


@pytest.mark.parametrize("I,O_,G,N", [(2, 64, 10, 1000), (64, 64, 10, 777), (64, 1, 10, 500), (5, 3, 5, 100)])
def test_layer_without_standalone_scaler(I, O_, G, N):
    """enable_standalone_scale_spline=False (kan.py:16, :145-151: the spline weight is used as it is; the ABI gets
    spline_scaler = NULL) through every layer kernel family: plain forward/backward and the 4-jet pair."""
    from pde_amd.kan import KANLinear
    torch.manual_seed(I * 7 + O_)
    lay = KANLinear(I, O_, grid_size=G, enable_standalone_scale_spline=False, scale_spline=0.7)
    assert not hasattr(lay, "spline_scaler")
    p64 = {"grid": lay.grid.double(), "base_weight": lay.base_weight.detach().double(),
           "spline_weight": lay.spline_weight.detach().double(), "spline_scaler": None}
    gen = torch.Generator().manual_seed(N)
    x = torch.rand(N, I, generator=gen) * 2.4 - 1.2
    gy = torch.randn(N, O_, generator=gen)
    ref = O.value_step(x.double(), [p64], "silu", gy.double())
    lay = lay.cuda()
    xg = x.cuda().requires_grad_(True)
    y = lay(xg)
    (y * gy.cuda()).sum().backward()
    assert max_rel(y.detach().cpu(), ref["y"]) <= TOL
    assert max_rel(xg.grad.cpu(), ref["gx"]) <= TOL
    assert max_rel(lay.base_weight.grad.cpu(), ref["grads"][0]["base_weight"]) <= TOL
    assert max_rel(lay.spline_weight.grad.cpu(), ref["grads"][0]["spline_weight"]) <= TOL
    # jets: d/dx_0, d/dx_1 and d2/dx_0^2 of every output against autograd on the oracle
    if I >= 2:
        xj = torch.zeros(N, I, 4)
        xj[:, :, 0] = x
        xj[:, 0, 1] = 1.0
        xj[:, 1, 2] = 1.0
        yj = lay.forward_jet(xj.cuda(), 2, 1).detach().cpu()
        xr = x.double().requires_grad_(True)
        yo = O.layer_forward(xr, p64, "silu")
        for o in range(min(O_, 3)):
            (g1,) = torch.autograd.grad(yo[:, o].sum(), xr, create_graph=True)
            (g2,) = torch.autograd.grad(g1[:, 0].sum(), xr, retain_graph=True)
            assert max_rel(yj[:, o, 1], g1[:, 0].detach()) <= TOL
            assert max_rel(yj[:, o, 2], g1[:, 1].detach()) <= TOL
            assert max_rel(yj[:, o, 3], g2[:, 0].detach()) <= TOL


def _sharded_oracle(x, layers, act, n, shard):
    loss, flat = 0.0, None
    for s0 in range(0, n, shard):
        r = O.pinn_step(x[s0:s0 + shard], layers, act, n_global=n)
        loss += r["loss"].item()
        f = _flat(r["grads"]).double()
        flat = f if flat is None else flat + f
        del r
    return loss, flat


def test_points_on_knots_cost_no_more_than_they_cost_the_reference_algorithm(record_property):
    """The benchmark configuration on 2^20 UNFILTERED seeded points.  About 670 of their 2^26 hidden activations land
    within 1e-6 of a knot of the next layer, where the third derivative of a cubic B-spline jumps (kan.py:94: the
    order-0 indicator is half-open), so any fp32 evaluation can put such a point's contribution on either side.
    Measured per tensor: err(HIP fp32 vs fp64 oracle) against err(fp32 oracle vs fp64 oracle), the reference
    algorithm's own fp32 loss on the same points.  The HIP step must not be further from fp64 than twice that
    (+1e-5)."""
    from pde_amd.pinn import BlackScholesStep
    widths, G, N, shard = [2, 64, 64, 1], 10, 1 << 20, 1 << 15
    m = _random_model(widths, G, 77)
    x = torch.rand(N, 2, generator=torch.Generator().manual_seed(1234)).cuda()
    sd = m.state_dict()
    lay32 = [{k: (v.cuda() if v is not None else None) for k, v in lay.items()} for lay in O.layers_from_state_dict(sd)]
    lay64 = [{k: (v.double() if v is not None else None) for k, v in lay.items()} for lay in lay32]
    loss64, flat64 = _sharded_oracle(x.double(), lay64, "silu", N, shard)
    loss32, flat32 = _sharded_oracle(x, lay32, "silu", N, shard)
    mg = m.cuda()
    step = BlackScholesStep(mg)
    loss = step(x).item()
    assert abs(loss - loss64) <= 2 * abs(loss32 - loss64) + TOL * abs(loss64)
    off = 0
    for l, lay in enumerate(mg.layers):
        for k in KEYS:
            n = getattr(lay, k).numel()
            e_hip = max_rel(step.flat_grad[off:off + n].double(), flat64[off:off + n])
            e_ref = max_rel(flat32[off:off + n], flat64[off:off + n])
            record_property(f"L{l}.{k}", f"hip {e_hip:.2e} ref32 {e_ref:.2e}")
            print(f"unfiltered 2^20: L{l}.{k}: HIP vs fp64 {e_hip:.3e}; fp32 oracle vs fp64 {e_ref:.3e}")
            assert e_hip <= 2 * e_ref + TOL, (l, k, e_hip, e_ref)
            # and the absolute bar on these seeded points (measured 1.4e-7 .. 4.6e-7 per tensor; another seed may put a
            # point on the other side of a jump of the third derivative: the relative bar above is the general one)
            assert e_hip <= 1e-5, (l, k, e_hip)
            off += n


def test_data_parallel_wrapper_on_a_one_rank_rccl_group():
    """DataParallelStep around the fused step with the nccl (RCCL) backend, world size 1: the in-place all-reduce
    of [flat_grad | loss] must hand back the plain step's bits, and replica sync must leave the parameters as
    they were."""
    import torch.distributed as dist
    from pde_amd.dist import DataParallelStep
    from pde_amd.pinn import BlackScholesStep
    m = _random_model([2, 64, 64, 1], 10, 21).cuda()
    x = torch.rand(5000, 2, generator=torch.Generator().manual_seed(22)).cuda()
    step = BlackScholesStep(m)
    step(x, n_global=3 * 5000)
    loss0, grad0 = step.loss.clone(), step.flat_grad.clone()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for sync in ("broadcast", "check"):
            dp = DataParallelStep(step, dist, sync=sync)
            loss = dp(x, n_global=3 * 5000)
            torch.cuda.synchronize()
            assert loss.data_ptr() == step.loss.data_ptr()          # reduced in place, no copy
            assert torch.equal(loss, loss0) and torch.equal(step.flat_grad, grad0)
        for k, v in m.state_dict().items():
            assert torch.equal(v, sd0[k])
    finally:
        if created:
            dist.destroy_process_group()


def test_jacrev_hessian_callers_stay_on_the_kernels_and_train(golden, monkeypatch):
    """run1.py:120-129 and :145-149 as written - jacrev / hessian through `forward`, residual, loss.backward() -
    on a GPU model with the tensor-code bases switched off: the dense-basis kernel (an elementwise op that carries
    its own derivative, vmap and forward-mode rules) must serve every transform level, and the parameter gradients
    that come back through those transforms must be the fused step's."""
    import pde_amd.kan as kan_mod
    from torch.func import hessian, jacrev
    from pde_amd.pinn import BS_RATE, BS_SIGMA, BlackScholesStep
    from test_gpu_parity import build_model

    def refuse(*a, **k):
        raise AssertionError("tensor-code b_splines used on a GPU tensor")

    monkeypatch.setattr(kan_mod, "dense_bsplines", refuse)
    rec = golden("jacrev_hessian")
    m = build_model(rec)
    x = torch.from_numpy(rec["x"][:, 0].copy()).cuda()
    t = torch.from_numpy(rec["x"][:, 1].copy()).cuda()

    def net(xx, tt):
        return m(torch.stack([xx.reshape(-1), tt.reshape(-1)]).permute(1, 0))[:, 0]

    s = net(x, t)
    s_x = jacrev(net, argnums=0)(x, t).sum(dim=0)
    s_xx = hessian(net, argnums=0)(x, t).sum(dim=0).sum(dim=0)
    s_t = jacrev(net, argnums=1)(x, t).sum(dim=0)
    for val, key in ((s, "s"), (s_x, "s_x"), (s_xx, "s_xx"), (s_t, "s_t")):
        assert max_rel(val.detach().cpu(), rec[key]) <= TOL, key
    res = s_t - 0.5 * BS_SIGMA ** 2 * x ** 2 * s_xx - BS_RATE * x * s_x + BS_RATE * s          # run1.py:128
    loss = torch.mean(res ** 2)                                                                 # run1.py:148
    for p in m.parameters():
        p.grad = None
    loss.backward()                                                                             # run1.py:225
    got = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    step = BlackScholesStep(m)
    loss_fused = step(torch.stack([x, t], 1))
    assert abs(loss.item() - loss_fused.item()) <= TOL * abs(loss_fused.item())
    assert max_rel(got.cpu(), step.flat_grad.cpu()) <= TOL
    # the module's b_splines method on a GPU tensor is the same kernel; against the tensor code on the CPU
    monkeypatch.undo()
    lay = m.layers[0]
    xs = torch.rand(300, lay.in_features, generator=torch.Generator().manual_seed(1)) * 3.4 - 1.7
    from pde_amd.spline_host import dense_bsplines
    assert max_rel(lay.b_splines(xs.cuda()).cpu(), dense_bsplines(xs, lay.grid.cpu(), 3)) <= 2e-6


def test_nested_autograd_grad_through_forward(golden):
    """The other way of forming run1.py:122-123's derivatives: nested torch.autograd.grad(create_graph=True) through
    `forward` (the formulation the fixtures themselves were made with on the reference KAN).  The first-order
    backward of the layer op is then recorded, so it is formed from the dense-basis kernel; values, s_x, s_t, s_xx and
    the parameter gradients of the residual loss must match the fixture / the fused step."""
    from pde_amd.pinn import BS_RATE, BS_SIGMA, BlackScholesStep
    from test_gpu_parity import build_model
    rec = golden("jacrev_hessian")
    m = build_model(rec)
    xt = torch.from_numpy(rec["x"].copy()).cuda().requires_grad_(True)
    s = m(xt)[:, 0]
    (g1,) = torch.autograd.grad(s.sum(), xt, create_graph=True)
    (g2,) = torch.autograd.grad(g1[:, 0].sum(), xt, create_graph=True)
    s_x, s_t, s_xx = g1[:, 0], g1[:, 1], g2[:, 0]
    for val, key in ((s, "s"), (s_x, "s_x"), (s_xx, "s_xx"), (s_t, "s_t")):
        assert max_rel(val.detach().cpu(), rec[key]) <= TOL, key
    x = xt[:, 0]
    res = s_t - 0.5 * BS_SIGMA ** 2 * x ** 2 * s_xx - BS_RATE * x * s_x + BS_RATE * s
    loss = torch.mean(res ** 2)
    for p in m.parameters():
        p.grad = None
    loss.backward()
    got = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    step = BlackScholesStep(m)
    loss_fused = step(xt.detach())
    assert abs(loss.item() - loss_fused.item()) <= TOL * abs(loss_fused.item())
    assert max_rel(got.cpu(), step.flat_grad.cpu()) <= TOL
    # jets refuse a recorded backward with a pointer to forward_jet instead of returning constants
    jet = m.forward_jet(xt.detach())
    with pytest.raises(RuntimeError, match="forward_jet"):
        torch.autograd.grad(jet.sum(), list(m.parameters()), create_graph=True)


def test_float64_model_on_the_gpu_under_jacrev_takes_the_tensor_code(golden):
    """The kernels are float32: a float64 model on the GPU under a torch.func transform must compute through the
    tensor-code bases (as it would on the CPU) instead of raising."""
    from torch.func import jacrev
    from test_gpu_parity import build_model
    rec = golden("jacrev_hessian")
    m = build_model(rec).double()
    x = torch.from_numpy(rec["x"][:, 0].copy()).cuda().double()
    t = torch.from_numpy(rec["x"][:, 1].copy()).cuda().double()

    def net(xx, tt):
        return m(torch.stack([xx.reshape(-1), tt.reshape(-1)]).permute(1, 0))[:, 0]

    s_x = jacrev(net, argnums=0)(x, t).sum(dim=0)
    assert s_x.dtype == torch.float64
    assert max_rel(s_x.cpu(), rec["s_x"]) <= TOL


def test_workspace_budget_picks_a_micro_batch_that_fits():
    """A byte budget instead of a micro-batch: the step takes the largest micro-batch whose workspace fits, and the
    gradients equal those of the unbounded step up to summation order."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model([2, 64, 64, 1], 10, 5).cuda()
    x = torch.rand(20000, 2, generator=torch.Generator().manual_seed(3)).cuda()
    full = BlackScholesStep(m)
    loss_full = full(x).item()
    g_full = full.flat_grad.clone()
    whole = full._ws.numel()
    bounded = BlackScholesStep(m, workspace_budget=whole // 3)
    loss_b = bounded(x).item()
    assert 0 < bounded.micro_batch < 20000 and bounded.micro_batch % 4096 == 0
    assert bounded._ws.numel() <= whole // 3
    assert abs(loss_b - loss_full) <= 1e-6 * abs(loss_full)
    assert max_rel(bounded.flat_grad.cpu(), g_full.cpu()) <= 3e-6
    with pytest.raises(Exception, match="budget"):
        BlackScholesStep(m, workspace_budget=1 << 16)(x)


@pytest.mark.parametrize("k", [1, 2, 4, 5])
def test_spline_orders_other_than_cubic(k):
    """`spline_order` is a constructor argument of the mirrored module (kan.py:12).  The whole-layer kernels are cubic;
    other orders run `forward` as its defining expression (kan.py:153-166) around the library's any-order basis kernel,
    which carries its own derivative: bases and their first two derivatives against autograd through the fp64
    oracle's Cox-de Boor recursion (kan.py:78-111), then a layer's forward and parameter / input gradients."""
    from pde_amd import functional as K
    from pde_amd.kan import KANLinear
    torch.manual_seed(k)
    I, O_, G, N = 7, 5, 6, 400
    lay = KANLinear(I, O_, grid_size=G, spline_order=k).cuda()
    with torch.no_grad():
        lay.grid[:, 3] += 0.03                                            # not equally spaced
        for p in lay.parameters():
            p.add_(0.1 * torch.randn_like(p))
    x = (torch.rand(N, I, generator=torch.Generator().manual_seed(2)) * 2.6 - 1.3).cuda()   # also off the base grid
    x64 = x.double().cpu().requires_grad_(True)
    g64 = lay.grid.double().cpu()
    b64 = O.b_splines(x64, g64, k)
    assert b64.shape == (N, I, G + k)
    got = K.basis_dense(x, lay.grid, G, k)
    assert max_rel(got.cpu(), b64.detach()) <= 2e-6
    cur = b64
    for order in (1, 2):
        # derivative of every basis with respect to its own input: one backward pass per basis index
        if order > k:                                                     # beyond the order: piecewise zero
            cur = torch.zeros_like(b64)
        else:
            cols = [torch.autograd.grad(cur[:, :, c].sum(), x64, create_graph=True)[0] for c in range(G + k)]
            cur = torch.stack(cols, dim=-1)
        got = K.basis_dense(x, lay.grid, G, k, order)
        scale = cur.detach().abs().max().clamp_min(1e-30)
        assert ((got.cpu().double() - cur.detach()).abs().max() / scale).item() <= 5e-6, order
    # the layer
    xg = x.clone().requires_grad_(True)
    y = lay(xg)
    w = torch.randn(N, O_, generator=torch.Generator().manual_seed(3)).cuda()
    (y * w).sum().backward()
    p64 = {n_: (p.detach().double().cpu().requires_grad_(True)) for n_, p in lay.named_parameters()}
    q = {"grid": g64, "base_weight": p64["base_weight"], "spline_weight": p64["spline_weight"],
         "spline_scaler": p64["spline_scaler"]}
    xr = x.double().cpu().requires_grad_(True)
    yr = O.layer_forward(xr, q, "silu", spline_order=k)
    (yr * w.double().cpu()).sum().backward()
    assert max_rel(y.detach().cpu(), yr.detach()) <= TOL
    assert max_rel(xg.grad.cpu(), xr.grad) <= TOL
    for n_, p in lay.named_parameters():
        assert max_rel(p.grad.cpu(), p64[n_].grad) <= TOL, n_


@pytest.mark.parametrize("widths,N", [([2, 64, 64, 1], 1), ([2, 64, 64, 1], 31), ([2, 64, 64, 1], 33),
                                      ([2, 128, 64, 1], 4097), ([2, 64, 1], 100001)])
def test_first_layer_kernel_for_two_inputs(widths, N, monkeypatch):
    """A 2-input layer with window tables takes a kernel of its own (resident blocks, batches of 32 points, stores
    straight from registers).  Batches with dead points, a single point, 128 outputs (two block columns) and more
    batches than resident waves: same step as with that kernel switched off, and as the fp64 oracle."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model(widths, 10, 21)
    x = torch.rand(N, 2, generator=torch.Generator().manual_seed(N))
    ref64 = O.pinn_step(x.double(), O.layers_from_state_dict(m.state_dict(), dtype=torch.float64), "silu")
    m = m.cuda()
    step = BlackScholesStep(m)
    res = torch.empty(N, device="cuda")
    loss = step(x.cuda(), residual_out=res).item()
    grad = step.flat_grad.clone()
    monkeypatch.setenv("PDEKAN_DISABLE_FEW_INPUTS", "1")
    res_b = torch.empty(N, device="cuda")
    loss_b = step(x.cuda(), residual_out=res_b).item()
    monkeypatch.delenv("PDEKAN_DISABLE_FEW_INPUTS")
    assert max_rel(res.cpu(), res_b.cpu()) <= 2e-6
    assert abs(loss - loss_b) <= 2e-6 * abs(loss_b)
    assert max_rel(grad.cpu(), step.flat_grad.cpu()) <= 5e-6
    flat64 = torch.cat([ref64["grads"][l][k].flatten() for l in range(len(widths) - 1) for k in KEYS])
    assert max_rel(res.cpu(), ref64["res"]) <= TOL
    assert abs(loss - ref64["loss"].item()) <= TOL * abs(ref64["loss"].item())
    assert max_rel(grad.cpu(), flat64) <= TOL


@pytest.mark.parametrize("widths,G,N,act", [([2, 5, 1], 5, 10000, torch.nn.SiLU), ([2, 6, 1], 5, 100, torch.nn.Identity),
                                            ([2, 4, 3, 1], 10, 777, torch.nn.SiLU), ([2, 1], 20, 65, torch.nn.SiLU),
                                            ([2, 8, 1], 5, 1, torch.nn.Identity), ([2, 3, 2, 2, 1], 5, 300, torch.nn.SiLU)])
def test_small_model_step_in_two_launches_equals_the_layer_by_layer_step(widths, G, N, act):
    """Models no wider than 8 take `small_step_kernel` (thread = point through every layer and back) and one
    reduction launch.  Same residuals, loss and gradients as the layer-by-layer kernels (switched on with
    PDEKAN_DISABLE_SMALL_STEP=1) and as the fp64 oracle; bitwise the same from run to run; with a target, a
    moved grid, a layer without its standalone scaler, and gradients accumulated into a second call."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model(widths, G, 31 + N % 7, act)
    if len(widths) > 2:
        m.layers[1].enable_standalone_scale_spline = False
        del m.layers[1].spline_scaler
    gen = torch.Generator().manual_seed(N)
    x = torch.rand(N, 2, generator=gen)
    target = 0.1 * torch.randn(N, generator=gen)
    m = m.cuda()
    if G == 10:                                           # knot rows that are no longer equally spaced
        with torch.no_grad():
            m.layers[0].update_grid(torch.rand(500, 2, generator=gen).cuda() * 1.2 - 0.1)
    name = "silu" if act is torch.nn.SiLU else "identity"
    layers64 = O.layers_from_state_dict({k: v.cpu() for k, v in m.state_dict().items()}, dtype=torch.float64)
    ref = O.pinn_step(x.double(), layers64, name, target=target.double())

    def run():
        step = BlackScholesStep(m)
        res = torch.empty(N, device="cuda")
        loss = step(x.cuda(), target=target.cuda(), residual_out=res)
        return loss.clone(), res, step.flat_grad.clone(), step

    old = os.environ.get("PDEKAN_DISABLE_SMALL_STEP")
    try:
        os.environ["PDEKAN_DISABLE_SMALL_STEP"] = "1"
        loss_l, res_l, grad_l, _ = run()
        os.environ["PDEKAN_DISABLE_SMALL_STEP"] = "0"
        loss_s, res_s, grad_s, step = run()
        loss_s2, res_s2, grad_s2, _ = run()
    finally:
        if old is None:
            os.environ.pop("PDEKAN_DISABLE_SMALL_STEP", None)
        else:
            os.environ["PDEKAN_DISABLE_SMALL_STEP"] = old
    assert torch.equal(loss_s, loss_s2) and torch.equal(res_s, res_s2) and torch.equal(grad_s, grad_s2)
    assert max_rel(res_s.cpu(), res_l.cpu()) <= 2e-6 and max_rel(grad_s.cpu(), grad_l.cpu()) <= 5e-6
    assert abs(loss_s.item() - loss_l.item()) <= 2e-6 * abs(loss_l.item())
    assert max_rel(res_s.cpu(), ref["res"]) <= TOL
    assert abs(loss_s.item() - ref["loss"].item()) <= TOL * abs(ref["loss"].item())
    flat_ref = torch.cat([ref["grads"][l][k].flatten() for l in range(len(widths) - 1) for k in KEYS
                          if ref["grads"][l][k] is not None])
    assert max_rel(grad_s.cpu(), flat_ref) <= TOL


@pytest.mark.parametrize("widths,N,moved,G", [([6, 64, 1], 1, False, 10), ([6, 64, 1], 77, True, 10),
                                              ([6, 128, 64, 1], 4101, False, 10), ([6, 256, 1], 300, False, 10),
                                              ([6, 128, 128, 1], 20000, True, 10), ([6, 64, 1], 1234, False, 5),
                                              ([6, 128, 1], 999, False, 20), ([6, 64, 1], 515, True, 20)])
def test_first_layer_kernels_of_the_basket_model(widths, N, moved, G, monkeypatch):
    """The first layer of a 12-jet model (6 inputs = 6 directions: its input jets are seeds) has a forward and a
    weight-gradient kernel of its own: 85 multiply-adds per output jet, weights in LDS, weight gradients summed in
    registers per chunk.  One point, batches with dead points, 64 to 256 outputs, grid sizes 5, 10 and 20 (a kernel
    each), a first layer whose knots update_grid has moved, a target: same step as with those kernels switched off
    (PDEKAN_DISABLE_SEEDED_FIRST=1: the streaming kernels), as the fp64 oracle, and bit for bit the same from run to
    run."""
    from pde_amd.pinn import BlackScholesStep
    m = _random_model(widths, G, 41 + N % 5)
    gen = torch.Generator().manual_seed(N)
    x = torch.rand(N, 6, generator=gen)
    target = 0.1 * torch.randn(N, generator=gen)
    m = m.cuda()
    if moved:
        with torch.no_grad():
            m.layers[0].update_grid(torch.rand(400, 6, generator=gen).cuda() * 1.2 - 0.1)
    layers64 = O.layers_from_state_dict({k: v.cpu() for k, v in m.state_dict().items()}, dtype=torch.float64)
    ref = O.pinn_step(x.double(), layers64, "silu", target=target.double())

    def run():
        step = BlackScholesStep(m)
        res = torch.empty(N, device="cuda")
        loss = step(x.cuda(), target=target.cuda(), residual_out=res)
        return loss.clone(), res, step.flat_grad.clone()

    loss_a, res_a, grad_a = run()
    loss_a2, res_a2, grad_a2 = run()
    monkeypatch.setenv("PDEKAN_DISABLE_SEEDED_FIRST", "1")
    loss_b, res_b, grad_b = run()
    monkeypatch.delenv("PDEKAN_DISABLE_SEEDED_FIRST")
    assert torch.equal(loss_a, loss_a2) and torch.equal(res_a, res_a2) and torch.equal(grad_a, grad_a2)
    assert max_rel(res_a.cpu(), res_b.cpu()) <= 2e-6 and max_rel(grad_a.cpu(), grad_b.cpu()) <= 5e-6
    assert abs(loss_a.item() - loss_b.item()) <= 2e-6 * abs(loss_b.item())
    assert not torch.equal(grad_a, grad_b)                  # (the switch does change the kernels)
    flat_ref = torch.cat([ref["grads"][l][k].flatten() for l in range(len(widths) - 1) for k in KEYS])
    assert max_rel(res_a.cpu(), ref["res"]) <= TOL
    assert abs(loss_a.item() - ref["loss"].item()) <= TOL * abs(ref["loss"].item())
    assert max_rel(grad_a.cpu(), flat_ref) <= TOL
