"""pde_amd.deeponet.PI_DeepONet (HIP kernels through the C ABI) against the
fixtures made from the reference's KAN class under the reference formulation
(run1.py:75-149, 209-233)."""
import pytest
import torch

from conftest import DEEPONET_CASES, DEEPONET_MODELS, golden_state_dict, max_rel

pytestmark = pytest.mark.gpu
TOL = 1e-5


def build(rec):
    from pde_amd import KAN
    from pde_amd.deeponet import PI_DeepONet
    widths = {"model1": rec["w12"], "model2": rec["w12"], "model4": rec["w4"], "model5": rec["w5"]}
    models = {}
    for name in DEEPONET_MODELS:
        m = KAN([int(w) for w in widths[name]], grid_size=int(rec["grid_size"]), base_activation=torch.nn.Identity)
        m.load_state_dict(golden_state_dict(rec, prefix=f"sd.{name}."))
        models[name] = m
    net = PI_DeepONet(models["model1"], models["model2"], models["model4"], models["model5"],
                      sigma=float(rec["sigma"]), rate=float(rec["rate"]), bc_weight=float(rec["bc_weight"]))
    return net.cuda()


def tensors(rec):
    t = lambda k: torch.from_numpy(rec[k]).cuda()
    return t("u1"), t("u2"), (t("x_i"), t("t_i"), t("out_i")), (t("x_b"), t("t_b"), t("out_b"))


def check(val, rec, key, tol=TOL):
    r32, r64 = rec[key], rec[key.replace(".", "64.", 1) if "." in key else key + "64"]
    v = val.detach().cpu()
    assert max_rel(v, r64) <= tol, (key, max_rel(v, r64))
    assert max_rel(v, r32) <= tol + max_rel(r32, r64), (key, max_rel(v, r32))


@pytest.mark.parametrize("case", DEEPONET_CASES)
def test_losses_and_gradients_match_reference(golden, case):
    rec = golden(case)
    net = build(rec)
    u1, u2, bc, pde = tensors(rec)
    s_pred = net.operator_net(u1, u2, None, None, bc[0], bc[1])
    res = net.residual_net(u1, u2, None, None, pde[0], pde[1])
    bc_loss = net.loss_bcs(u1, u2, None, None, *bc)
    pde_loss = net.loss_res(u1, u2, None, None, *pde)
    loss = pde_loss + net.bc_weight * bc_loss
    loss.backward()
    check(s_pred, rec, "s_pred")
    check(res, rec, "res")
    check(bc_loss, rec, "bc_loss")
    check(pde_loss, rec, "pde_loss")
    check(loss, rec, "loss")
    for mname in DEEPONET_MODELS:
        for pname, p in getattr(net, mname).named_parameters():
            check(p.grad, rec, f"grad.{mname}.{pname}")


@pytest.mark.parametrize("case", DEEPONET_CASES)
def test_one_call_step_matches_reference(golden, case):
    """pdekan_deeponet_step (the whole step in four launches): predictions, residuals, the three losses and every
    gradient against the fixtures of the reference formulation."""
    from pde_amd.deeponet import DeepONetStep
    rec = golden(case)
    net = build(rec)
    u1, u2, bc, pde = tensors(rec)
    step = DeepONetStep(net)
    assert step.supported()
    s_pred, res = torch.empty_like(bc[0]), torch.empty_like(pde[0])
    losses = step(u1, u2, bc, pde, s_pred=s_pred, res_out=res)
    assert losses is not None and step.calls == 1
    check(s_pred, rec, "s_pred")
    check(res, rec, "res")
    check(losses[0], rec, "loss")
    check(losses[1], rec, "pde_loss")
    check(losses[2], rec, "bc_loss")
    for mname in DEEPONET_MODELS:
        for pname, p in getattr(net, mname).named_parameters():
            check(p.grad, rec, f"grad.{mname}.{pname}")
    first = [p.grad.clone() for p in net.parameters()]
    l0 = losses.clone()
    step(u1, u2, bc, pde)
    assert torch.equal(l0, step.losses)                          # fixed summation orders: the same bits every time
    assert all(torch.equal(a, p.grad) for a, p in zip(first, net.parameters()))


def _random_deeponet(widths12, widths4, widths5, grid, act, seed):
    from pde_amd import KAN
    from pde_amd.deeponet import PI_DeepONet
    torch.manual_seed(seed)
    mk = lambda w: KAN(w, grid_size=grid, base_activation=act)
    net = PI_DeepONet(mk(widths12), mk(widths12), mk(widths4), mk(widths5), bc_weight=8.0).cuda()
    with torch.no_grad():
        for p in net.parameters():                                # off the small default spline weights
            p.add_(0.3 * torch.randn_like(p))
    return net


@pytest.mark.parametrize("widths12,widths4,widths5,grid,act,m,n_bc,n_pde", [
    ([3, 2, 1], [1000, 2, 1], [2, 6, 1], 5, torch.nn.Identity, 1000, 100, 100),      # run1.py:478-482 at P = 3000
    ([3, 2, 1], [3000, 2, 1], [2, 6, 1], 5, torch.nn.Identity, 3000, 100, 100),      # the script's own sizes
    ([3, 4, 4, 1], [257, 4, 3, 2], [2, 8, 8, 2], 10, torch.nn.SiLU, 257, 33, 70),    # deeper, SiLU, ragged sizes
    ([2, 1], [64, 3], [2, 3], 20, torch.nn.SiLU, 64, 1, 300),                         # one-layer networks, grid 20
    ([3, 8, 8, 1], [40, 8, 8], [2, 8, 8, 8], 5, torch.nn.Identity, 40, 700, 5),       # 64 edges per layer
])
def test_one_call_step_agrees_with_the_layerwise_path(widths12, widths4, widths5, grid, act, m, n_bc, n_pde):
    """The one-call step against the same step run layer by layer under autograd (the path the fixtures pin), on
    networks and batch sizes the fixtures do not cover."""
    from pde_amd.deeponet import DeepONetStep
    net = _random_deeponet(widths12, widths4, widths5, grid, act, seed=m)
    g = torch.Generator().manual_seed(m + 1)
    r = lambda *shape: torch.rand(*shape, generator=g).cuda()
    u1, u2 = r(m, widths12[0]) * 1.6 - 0.8, r(m, widths12[0]) * 1.6 - 0.8
    bc = (r(n_bc) * 1.9 - 0.95, r(n_bc) * 1.9 - 0.95, r(n_bc))
    pde = (r(n_pde) * 1.9 - 0.95, r(n_pde) * 1.9 - 0.95, 0.1 * r(n_pde))
    bc_loss = net.loss_bcs(u1, u2, None, None, *bc)
    pde_loss = net.loss_res(u1, u2, None, None, *pde)
    loss = pde_loss + net.bc_weight * bc_loss
    loss.backward()
    want = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    step = DeepONetStep(net)
    losses = step(u1, u2, bc, pde)
    assert losses is not None
    assert max_rel(losses.cpu(), torch.stack([loss, pde_loss, bc_loss]).detach().cpu()) <= 2e-5
    for (name, p), w in zip(net.named_parameters(), want):
        assert max_rel(p.grad.cpu(), w.cpu()) <= 5e-5, (name, max_rel(p.grad.cpu(), w.cpu()))


def test_one_call_step_without_standalone_scalers_and_on_moved_grids():
    """Layers built with enable_standalone_scale_spline=False (kan.py:16: no spline_scaler parameter, the ABI gets
    NULL) and knot rows moved off their equal spacing (what update_grid leaves): the one-call step against the
    layer-by-layer path."""
    from pde_amd.deeponet import DeepONetStep
    from pde_amd.kan import KANLinear
    net = _random_deeponet([3, 3, 1], [90, 3, 2], [2, 5, 2], 5, torch.nn.SiLU, seed=9)
    for m in (net.model1, net.model2, net.model4, net.model5):   # (KAN's constructor has no switch for it, kan.py:240-252)
        m.layers = torch.nn.ModuleList(
            KANLinear(l.in_features, l.out_features, grid_size=5, base_activation=torch.nn.SiLU,
                      enable_standalone_scale_spline=False, scale_spline=0.6) for l in m.layers)
    net = net.cuda()
    g = torch.Generator().manual_seed(10)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.3 * torch.randn(p.shape, generator=g).cuda())
    with torch.no_grad():
        for m in (net.model1, net.model2, net.model4, net.model5):
            for layer in m.layers:
                layer.grid[:, 3:-3] += (0.12 * torch.rand(layer.grid[:, 3:-3].shape, generator=g) - 0.06).cuda()
    r = lambda *shape: torch.rand(*shape, generator=g).cuda()
    u1, u2 = r(90, 3) * 1.6 - 0.8, r(90, 3) * 1.6 - 0.8
    bc, pde = (r(50) * 1.8 - 0.9, r(50) * 1.8 - 0.9, r(50)), (r(61) * 1.8 - 0.9, r(61) * 1.8 - 0.9, 0.1 * r(61))
    bc_loss = net.loss_bcs(u1, u2, None, None, *bc)
    pde_loss = net.loss_res(u1, u2, None, None, *pde)
    (pde_loss + net.bc_weight * bc_loss).backward()
    want = [p.grad.clone() for p in net.parameters()]
    assert not any("spline_scaler" in n for n, _ in net.named_parameters())
    for p in net.parameters():
        p.grad = None
    losses = DeepONetStep(net)(u1, u2, bc, pde)
    assert losses is not None
    assert max_rel(losses[1:].cpu(), torch.stack([pde_loss, bc_loss]).detach().cpu()) <= 2e-5
    for (name, p), w in zip(net.named_parameters(), want):
        assert max_rel(p.grad.cpu(), w.cpu()) <= 5e-5, (name, max_rel(p.grad.cpu(), w.cpu()))


def test_one_call_step_refuses_what_it_cannot_hold(monkeypatch):
    """Networks wider than the kernels hold, and the A/B switch: train_step then runs layer by layer, same numbers."""
    from pde_amd.deeponet import DeepONetStep
    net = _random_deeponet([3, 2, 1], [50, 2, 1], [2, 12, 1], 5, torch.nn.Identity, seed=3)     # 24 edges in a layer > 8 wide
    g = torch.Generator().manual_seed(5)
    r = lambda *shape: torch.rand(*shape, generator=g).cuda()
    u1, u2 = r(50, 3), r(50, 3)
    bc, pde = (r(20), r(20), r(20)), (r(20), r(20), r(20))
    assert DeepONetStep(net)(u1, u2, bc, pde) is None
    net2 = _random_deeponet([3, 2, 1], [50, 2, 1], [2, 6, 1], 5, torch.nn.Identity, seed=4)
    opt = torch.optim.AdamW(net2.parameters(), lr=0.01)
    a = net2.train_step(opt, u1, u2, None, None, bc, pde)
    assert net2._step.calls == 1
    monkeypatch.setenv("PDEKAN_DISABLE_DEEPONET_STEP", "1")
    net3 = _random_deeponet([3, 2, 1], [50, 2, 1], [2, 6, 1], 5, torch.nn.Identity, seed=4)
    opt3 = torch.optim.AdamW(net3.parameters(), lr=0.01)
    b = net3.train_step(opt3, u1, u2, None, None, bc, pde)
    assert net3._step.calls == 0
    assert max_rel(torch.stack(a).cpu(), torch.stack(b).cpu()) <= 2e-5
    for p, q in zip(net2.parameters(), net3.parameters()):
        assert max_rel(p.detach().cpu(), q.detach().cpu()) <= 2e-5


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("case", DEEPONET_CASES)
def test_adamw_steps_follow_the_reference(golden, case, fused):
    """Three AdamW(lr=0.01) steps of the loop body run1.py:217-233: the loss
    trace and one updated tensor; as one library call per step and layer by layer."""
    rec = golden(case)
    net = build(rec)
    net.fused = fused
    u1, u2, bc, pde = tensors(rec)
    opt = torch.optim.AdamW(net.parameters(), lr=0.01)
    trace = []
    for _ in range(len(rec["adamw_losses"])):
        loss, _, _ = net.train_step(opt, u1, u2, None, None, bc, pde)
        trace.append(loss.item())
    assert (net._step is not None and net._step.calls == len(trace)) == fused
    assert max_rel(torch.tensor(trace), rec["adamw_losses64"]) <= 2e-5
    w = net.model5.layers[0].spline_weight
    assert max_rel(w.detach().cpu(), rec["adamw_final64.model5.layers.0.spline_weight"]) <= 2e-5


def test_column_inputs_and_strided_points(golden):
    """The scripts hand x, t as [n] tensors from a DataLoader (run1.py:455-473) and
    as [n, 1] columns at evaluation (run1.py:495-503): same numbers either way."""
    rec = golden("deeponet_run1")
    net = build(rec)
    u1, u2, bc, _ = tensors(rec)
    a = net.operator_net(u1, u2, None, None, bc[0], bc[1])
    b = net.operator_net(u1, u2, None, None, bc[0].unsqueeze(1), bc[1].unsqueeze(1))
    assert torch.equal(a, b)
    j = net.operator_jet(u1, u2, None, None, bc[0], bc[1])
    assert j.shape == (bc[0].numel(), 4)
    assert torch.equal(j[:, 0], a)


@pytest.mark.parametrize("graph", [False, True])
def test_example_training_run_learns(graph):
    """examples/run1_pi_deeponet.py end to end at a reduced size: finite losses,
    and the boundary loss falls over the epochs; eagerly and with the step
    captured in a HIP graph and replayed."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "run1_pi_deeponet.py")
    spec = importlib.util.spec_from_file_location("run1_pi_deeponet", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(["--P", "300", "--Q", "200", "--batch", "100", "--epochs", "12", "--seed", "0"] +
                   (["--graph"] if graph else []))
    hist = torch.tensor(out["history"])
    assert torch.isfinite(hist).all()
    assert hist[-3:, 1].mean() < hist[:3, 1].mean()
    assert out["mse"] == out["mse"]
    # the script's evaluation on its 72 market quotes (run1.py:492-523): scored, finite
    assert out["market"]["n"] == 72 and out["market"]["mape"] == out["market"]["mape"]


def test_trained_pinn_prices_the_call():
    """examples/bs_pinn_accuracy.py at a reduced length: a KAN PINN trained with
    the fused step must land on the closed-form European call (B-S.py:5-17) and
    on the finite-difference grid (FDM.py:18-63) - an end-to-end check of the
    jets, the residual and the gradients against the physics."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "bs_pinn_accuracy.py")
    spec = importlib.util.spec_from_file_location("bs_pinn_accuracy", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import numpy as np
    from oracle.bs_accuracy import closed_form_call, fdm_call_grid
    out = mod.main(["--steps", "800", "--seed", "0"])
    assert out["history"][-1][1] < out["history"][1][1]        # PDE loss falls once the boundary data bite
    assert out["rmse_closed_form"] < 2e-2                      # prices range up to 1.27
    K, r, v = out["strike"], out["rate"], out["sigma"]
    exact = closed_form_call(out["S"], K, out["tau"], r, v)
    assert np.sqrt(np.mean((out["pred"] - exact) ** 2)) < 2e-2
    M = 400
    f = fdm_call_grid(1.0, r, v, M, K, fold_boundary=True)
    ti = np.rint((1.0 - out["tau"]) * M).astype(int)           # row i = time level; time to expiry = 1 - i/M
    si = np.rint(out["S"] / (3.0 * K) * M).astype(int)
    assert np.sqrt(np.mean((out["pred"] - f[ti, si]) ** 2)) < 2e-2


@pytest.mark.parametrize("widths12,widths4,widths5,m,n_bc,n_pde", [
    ([3, 2, 1], [3000, 2, 1], [2, 6, 1], 3000, 100, 100),
    ([3, 8, 8, 1], [37, 8, 8], [2, 8, 8, 8], 37, 1030, 3),          # more trunk blocks than the cap of 64
])
def test_one_call_step_stays_inside_its_workspace(widths12, widths4, widths5, m, n_bc, n_pde):
    """pdekan_deeponet_step given exactly pdekan_deeponet_workspace_bytes() inside a larger allocation whose
    remainder holds a pattern: the remainder survives, the numbers equal the ordinary call's, less is refused;
    without gradient pointers the call writes the losses only."""
    from pde_amd import _lib, functional as K
    from pde_amd.deeponet import DeepONetStep
    lib = _lib.load()
    net = _random_deeponet(widths12, widths4, widths5, 5, torch.nn.Identity, seed=m)
    g = torch.Generator().manual_seed(m)
    r = lambda *shape: torch.rand(*shape, generator=g).cuda()
    u1, u2 = r(m, 3), r(m, 3)
    bc, pde = (r(n_bc), r(n_bc), r(n_bc)), (r(n_pde), r(n_pde), r(n_pde))
    step = DeepONetStep(net)
    assert step(u1, u2, bc, pde) is not None
    loss0, grad0 = step.losses.clone(), step.flat_grad.clone()
    nets = step._structs()
    need = lib.pdekan_deeponet_workspace_bytes(nets, m, n_bc, n_pde)
    assert need > 0
    guard = 1 << 16
    buf = torch.empty(need + guard, dtype=torch.uint8, device="cuda")
    buf[need:] = 0xA5
    step.flat_grad.zero_()
    step.losses.zero_()
    args = (K._ptr(u1), K._ptr(u2), m, K._ptr(bc[0]), K._ptr(bc[1]), K._ptr(bc[2]), n_bc, K._ptr(pde[0]), K._ptr(pde[1]),
            K._ptr(pde[2]), n_pde, net.sigma, net.rate, net.bc_weight, K._ptr(step.losses), None, None)
    _lib.check(lib.pdekan_deeponet_step(nets, *args, K._ptr(buf), need, K._stream(u1.device)))
    torch.cuda.synchronize()
    assert bool((buf[need:] == 0xA5).all()), "a kernel wrote past the end of the workspace"
    assert torch.equal(step.losses, loss0) and torch.equal(step.flat_grad, grad0)
    assert lib.pdekan_deeponet_step(nets, *args, K._ptr(buf), need - 257, K._stream(u1.device)) == -2
    # losses only
    for k in range(4):
        nets[k].grads = None
    step.flat_grad.fill_(7.0)
    step.losses.zero_()
    _lib.check(lib.pdekan_deeponet_step(nets, *args, K._ptr(buf), need, K._stream(u1.device)))
    torch.cuda.synchronize()
    assert torch.equal(step.losses, loss0) and bool((step.flat_grad == 7.0).all())
    # a sensor count that is not model4's input width is refused with a message
    assert lib.pdekan_deeponet_workspace_bytes(nets, m + 1, n_bc, n_pde) == 0
    assert b"sensor" in lib.pdekan_last_error()
