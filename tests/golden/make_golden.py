"""Generate the golden fixtures under tests/golden/ from the reference itself.

Run in the build container only (it needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's ``efficient_kan`` package, builds seeded models at
the benchmark shapes (the two widest narrowed so the files stay small),
evaluates forward values, per-point first/second derivatives, the Black-Scholes
residual, the mean-squared loss and every parameter gradient in fp32 and in
fp64, and writes one ``.npz`` per case.  The fixtures hold inputs and outputs
only - no reference source text.

One extra case (``jacrev_hessian``) evaluates the derivatives exactly as
run1.py:122-124 does (``torch.func.jacrev`` / ``hessian`` followed by sums) to
pin that formulation to the per-point one used everywhere else.
"""
import copy
import os
import sys

import numpy as np
import torch

REF = os.environ.get("PDE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
from efficient_kan import KAN  # noqa: E402  (the reference package)

OUT = os.path.dirname(os.path.abspath(__file__))
SIGMA, RATE = 0.165856529, 0.025610   # run1.py:126-127


def special_points(d, grid_size, n_rand, gen):
    """Uniform [0,1)^d points plus points on knots, at the ends of the
    extended grid and outside it (first column only; the others stay random)."""
    x = torch.rand(n_rand, d, generator=gen)
    h = 2.0 / grid_size
    knots = torch.arange(-3, grid_size + 4) * h - 1.0
    extra = torch.rand(knots.numel() + 6, d, generator=gen)
    extra[: knots.numel(), 0] = knots.float()
    extra[knots.numel():, 0] = torch.tensor([-5.0, 5.0, -1.0 - 3 * h - 1e-3, 1.0 + 3 * h + 1e-3,
                                             0.999999, -0.999999])
    return torch.cat([x, extra], 0)


def jets_autograd(model, x, n_second):
    xr = x.detach().clone().requires_grad_(True)
    v = model(xr)[:, 0]
    (g1,) = torch.autograd.grad(v.sum(), xr, create_graph=True)
    sec = []
    for j in range(n_second):
        (g2,) = torch.autograd.grad(g1[:, j].sum(), xr, create_graph=True)
        sec.append(g2[:, j])
    sec = torch.stack(sec, 1) if sec else xr.new_zeros(x.size(0), 0)
    return xr, v, g1, sec


def residual(xr, v, g1, sec):
    d = xr.size(1)
    res = g1[:, d - 1] + RATE * v
    for j in range(d - 1):
        res = res - 0.5 * SIGMA ** 2 * xr[:, j] ** 2 * sec[:, j] - RATE * xr[:, j] * g1[:, j]
    return res


def run_case(model, x, tag):
    out = {}
    n_second = x.size(1) - 1
    model.zero_grad()
    xr, v, g1, sec = jets_autograd(model, x, n_second)
    res = residual(xr, v, g1, sec)
    loss = (res ** 2).mean()
    loss.backward()
    out[f"v{tag}"] = v.detach().numpy()
    out[f"first{tag}"] = g1.detach().numpy()
    out[f"second{tag}"] = sec.detach().numpy()
    out[f"res{tag}"] = res.detach().numpy()
    out[f"loss{tag}"] = loss.detach().numpy()
    for name, p in model.named_parameters():
        out[f"grad{tag}.{name}"] = p.grad.detach().numpy().copy()
    # plain forward + first-order backward with a fixed cotangent
    model.zero_grad()
    xv = x.detach().clone().requires_grad_(True)
    y = model(xv)
    gy = torch.linspace(-1.0, 1.0, y.numel(), dtype=y.dtype).view_as(y)
    (y * gy).sum().backward()
    out[f"y{tag}"] = y.detach().numpy()
    out[f"gx{tag}"] = xv.grad.detach().numpy()
    for name, p in model.named_parameters():
        out[f"vgrad{tag}.{name}"] = p.grad.detach().numpy().copy()
    return out


def make(name, widths, grid_size, act, seed, n_rand, perturb=True, update_grid=False):
    torch.manual_seed(seed)
    act_cls = {"silu": torch.nn.SiLU, "identity": torch.nn.Identity}[act]
    model = KAN(widths, grid_size=grid_size, base_activation=act_cls)
    init_sd = {k: v.clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(seed + 1000)
    if perturb:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
    x = special_points(widths[0], grid_size, n_rand, gen)
    if update_grid:
        with torch.no_grad():
            model(torch.rand(256, widths[0], generator=gen) * 1.4 - 0.2, update_grid=True)
    rec = {
        "widths": np.array(widths), "grid_size": np.array(grid_size), "act": np.array(act),
        "seed": np.array(seed), "x": x.numpy(),
    }
    for k, v in init_sd.items():
        rec[f"init.{k}"] = v.numpy()
    for k, v in model.state_dict().items():
        rec[f"sd.{k}"] = v.numpy()
    rec.update(run_case(model, x, ""))
    m64 = copy.deepcopy(model).double()
    rec.update(run_case(m64, x.double(), "64"))
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: N={x.size(0)} loss={float(rec['loss']):.6e} -> {os.path.getsize(path)/1024:.0f} KiB")


def make_jacrev_case():
    """Derivatives the way the scripts take them (run1.py:121-124)."""
    from torch.func import hessian, jacrev
    torch.manual_seed(7)
    model = KAN([2, 6, 1], grid_size=5, base_activation=torch.nn.Identity)
    gen = torch.Generator().manual_seed(8)
    x = torch.rand(24, generator=gen)
    t = torch.rand(24, generator=gen)

    def net(xx, tt):
        y = torch.stack([xx.reshape(-1), tt.reshape(-1)]).permute(1, 0)   # run1.py:101-106
        return model(y)[:, 0]

    s = net(x, t)
    s_x = jacrev(net, argnums=0)(x, t).sum(dim=0)
    s_xx = hessian(net, argnums=0)(x, t).sum(dim=0).sum(dim=0)
    s_t = jacrev(net, argnums=1)(x, t).sum(dim=0)
    res = s_t - 0.5 * SIGMA ** 2 * x ** 2 * s_xx - RATE * x * s_x + RATE * s
    rec = {"x": torch.stack([x, t], 1).numpy(), "s": s.detach().numpy(), "s_x": s_x.detach().numpy(),
           "s_xx": s_xx.detach().numpy(), "s_t": s_t.detach().numpy(), "res": res.detach().numpy(),
           "widths": np.array([2, 6, 1]), "grid_size": np.array(5), "act": np.array("identity")}
    for k, v in model.state_dict().items():
        rec[f"sd.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "jacrev_hessian.npz"), **rec)
    print("jacrev_hessian: ok")


def make_deeponet(name, sensors, w12, w4, w5, seed, n, bc_weight, steps=3):
    """The physics-informed DeepONet of run1.py:75-149 on the reference's KAN class:
    model1(u1)*model2(u2) flattened to one row, repeated n times into model4
    (run1.py:109-113), trunk model5 on (x, t), sum of branch*trunk (:116);
    derivatives by jacrev/hessian (:121-124); loss = pde + w*bc (:224); then
    `steps` AdamW(lr=0.01) updates (:209, :217-233) with the losses recorded."""
    from torch.func import hessian, jacrev
    torch.manual_seed(seed)
    ident = torch.nn.Identity
    models = {"model1": KAN(w12, base_activation=ident), "model2": KAN(w12, base_activation=ident),
              "model4": KAN(w4, base_activation=ident), "model5": KAN(w5, base_activation=ident)}
    gen = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for m in models.values():
            for p in m.parameters():
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
    u1 = torch.rand(sensors, w12[0], generator=gen)
    u2 = torch.rand(sensors, w12[0], generator=gen)
    x_i, t_i, out_i = torch.rand(n, generator=gen), torch.rand(n, generator=gen), torch.rand(n, generator=gen)
    x_b, t_b, out_b = torch.rand(n, generator=gen), torch.rand(n, generator=gen), torch.zeros(n)

    def evaluate(ms, cast, tag, rec):
        c = lambda z: z.to(cast)

        def net(xx, tt):
            k = len(xx)
            rows = (ms["model1"](c(u1)) * ms["model2"](c(u2))).view(1, -1).repeat(k, 1)
            y = torch.stack([xx.reshape(-1), tt.reshape(-1)]).permute(1, 0)
            return torch.sum(ms["model4"](rows) * ms["model5"](y), dim=1)

        def all_losses():
            s_pred = net(c(x_i), c(t_i))
            bc_loss = torch.mean((c(out_i).flatten() - s_pred) ** 2)
            xb, tb = c(x_b), c(t_b)
            s = net(xb, tb)
            s_x = jacrev(net, argnums=0)(xb, tb).sum(dim=0)
            s_xx = hessian(net, argnums=0)(xb, tb).sum(dim=0).sum(dim=0)
            s_t = jacrev(net, argnums=1)(xb, tb).sum(dim=0)
            res = s_t - 0.5 * SIGMA ** 2 * xb ** 2 * s_xx - RATE * xb * s_x + RATE * s
            pde_loss = torch.mean((c(out_b).flatten() - res) ** 2)
            return s_pred, res, bc_loss, pde_loss, pde_loss + bc_weight * bc_loss

        for m in ms.values():
            m.zero_grad()
        s_pred, res, bc_loss, pde_loss, loss = all_losses()
        loss.backward()
        rec[f"s_pred{tag}"] = s_pred.detach().numpy()
        rec[f"res{tag}"] = res.detach().numpy()
        rec[f"bc_loss{tag}"] = bc_loss.detach().numpy()
        rec[f"pde_loss{tag}"] = pde_loss.detach().numpy()
        rec[f"loss{tag}"] = loss.detach().numpy()
        for mname, m in ms.items():
            for pname, p in m.named_parameters():
                rec[f"grad{tag}.{mname}.{pname}"] = p.grad.detach().numpy().copy()
        params = [p for m in ms.values() for p in m.parameters()]
        opt = torch.optim.AdamW(params, lr=0.01)
        trace = []
        for _ in range(steps):
            opt.zero_grad()
            _, _, _, _, loss = all_losses()
            loss.backward()
            opt.step()
            trace.append(float(loss.detach()))
        rec[f"adamw_losses{tag}"] = np.array(trace)
        rec[f"adamw_final{tag}.model5.layers.0.spline_weight"] = ms["model5"].layers[0].spline_weight.detach().numpy().copy()

    rec = {"sensors": np.array(sensors), "w12": np.array(w12), "w4": np.array(w4), "w5": np.array(w5),
           "grid_size": np.array(5), "act": np.array("identity"), "bc_weight": np.array(float(bc_weight)),
           "sigma": np.array(SIGMA), "rate": np.array(RATE),
           "u1": u1.numpy(), "u2": u2.numpy(), "x_i": x_i.numpy(), "t_i": t_i.numpy(), "out_i": out_i.numpy(),
           "x_b": x_b.numpy(), "t_b": t_b.numpy(), "out_b": out_b.numpy()}
    for mname, m in models.items():
        for k, v in m.state_dict().items():
            rec[f"sd.{mname}.{k}"] = v.numpy().copy()
    m64 = {k: copy.deepcopy(m).double() for k, m in models.items()}
    evaluate(models, torch.float32, "", rec)
    evaluate(m64, torch.float64, "64", rec)
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: loss={float(rec['loss']):.6e} adamw={rec['adamw_losses']} -> {os.path.getsize(path)/1024:.0f} KiB")


def make_lbfgs_trace(name="lbfgs_trace", outer_steps=3, max_iter=6):
    """The LBFGS training variant (run2.py:174-199, 测试.py:124): the reference KAN driven by the reference's own
    vendored optimiser (kan/LBFGS.py:185-483, loaded from its file: the `kan` package around it needs libraries
    this container lacks) with the Black-Scholes residual loss as the closure, a few outer steps on one fixed batch.
    Stored: the initial state, the batch, the loss of EVERY closure evaluation in order, and the parameters after
    each outer step."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_vendored_lbfgs", os.path.join(REF, "kan", "LBFGS.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(21)
    widths, grid_size = [2, 8, 1], 5
    model = KAN(widths, grid_size=grid_size)
    gen = torch.Generator().manual_seed(22)
    x = torch.rand(512, 2, generator=gen)
    rec = {"widths": np.array(widths), "grid_size": np.array(grid_size), "act": np.array("silu"), "x": x.numpy(),
           "outer_steps": np.array(outer_steps), "max_iter": np.array(max_iter)}
    for k, v in model.state_dict().items():
        rec[f"sd.{k}"] = v.detach().clone().numpy()
    opt = mod.LBFGS(model.parameters(), lr=1.0, max_iter=max_iter, history_size=10, line_search_fn="strong_wolfe",
                    tolerance_grad=1e-32, tolerance_change=1e-32, tolerance_ys=1e-32)
    evals = []

    def closure():
        opt.zero_grad()
        xr, v, g1, sec = jets_autograd(model, x, 1)
        loss = (residual(xr, v, g1, sec) ** 2).mean()
        loss.backward()
        evals.append(float(loss))
        return loss

    counts = []
    for k in range(outer_steps):
        opt.step(closure)
        counts.append(len(evals))
        for name_, v in model.state_dict().items():
            rec[f"after{k}.{name_}"] = v.detach().clone().numpy()
    rec["closure_losses"] = np.array(evals, np.float64)
    rec["evals_after_step"] = np.array(counts)
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: {len(evals)} closure evaluations, loss {evals[0]:.4e} -> {evals[-1]:.4e}")


def make_data_quotes(name="data_quotes"):
    """The 72 market quotes the scripts score against (data.csv: K, X, T, V; run1.py:492-507) as arrays."""
    rows = np.loadtxt(os.path.join(REF, "data.csv"), delimiter=",", skiprows=1)
    assert rows.shape == (72, 4)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), K=rows[:, 0], X=rows[:, 1], T=rows[:, 2], V=rows[:, 3])
    print(f"{name}: {rows.shape[0]} quotes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--extras":      # only the fixtures added later
        make_lbfgs_trace()
        make_data_quotes()
        sys.exit(0)
    make("cfg1_silu", [2, 5, 1], 5, "silu", 0, 200)
    make("cfg1_identity", [2, 5, 1], 5, "identity", 1, 200)
    make("cfg1_init", [2, 5, 1], 5, "silu", 0, 64, perturb=False)
    make("cfg2_silu", [2, 64, 64, 1], 10, "silu", 2, 256)
    make("cfg2_identity", [2, 64, 64, 1], 10, "identity", 3, 128)
    make("cfg4_narrow", [2, 32, 32, 32, 1], 20, "silu", 4, 128)
    make("cfg5_narrow", [6, 16, 16, 1], 10, "silu", 5, 128)
    make("updated_grid", [2, 8, 1], 5, "silu", 6, 128, update_grid=True)
    make("deeponet_trunk", [2, 6, 1], 5, "identity", 9, 100)
    make_jacrev_case()
    make_deeponet("deeponet_run1", 12, [3, 2, 1], [12, 2, 1], [2, 6, 1], 11, 16, 8.0)      # shapes of run1.py:478-482
    make_deeponet("deeponet_wide", 10, [3, 2, 1], [10, 2, 2], [2, 5, 2], 12, 20, 200.0)    # latent width 2 (the 3-branch script's model4)
    make_lbfgs_trace()
    make_data_quotes()
