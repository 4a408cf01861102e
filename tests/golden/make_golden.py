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


if __name__ == "__main__":
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
