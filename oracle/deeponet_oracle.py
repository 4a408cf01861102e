"""CPU restatement of the physics-informed DeepONet the reference scripts wrap
around their KANs (run1.py:75-149; same structure in run2.py:86-160 and
run3.py:87-161).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else) as the checker
for pde_amd/deeponet.py.  Parity PINNED: tests/test_oracle_golden.py checks
these functions against tests/golden/deeponet_*.npz, which
tests/golden/make_golden.py produced by running the reference's own KAN class
through the reference formulation (torch.func.jacrev / hessian).

The networks are passed as callables ``x -> y`` so the same code runs on the
oracle's layer functions (kan_oracle.model_forward) in any dtype.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence

import torch
from torch.func import hessian, jacrev

from . import kan_oracle as K

Net = Callable[[torch.Tensor], torch.Tensor]


def stack_xt(x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """run1.py:101-106: flatten both, stack, permute -> [n, 2] (non-contiguous)."""
    return torch.stack([x.reshape(-1), t.reshape(-1)]).permute(1, 0)


def operator_net(nets: Dict[str, Net], u1: torch.Tensor, u2: torch.Tensor, x: torch.Tensor, t: torch.Tensor):
    """run1.py:109-117.  The sensor product is flattened to ONE row, repeated n
    times, pushed through model4; the trunk sees (x, t); the output is the sum
    over the latent width of branch * trunk."""
    n = len(x)
    prod = nets["model1"](u1) * nets["model2"](u2)              # run1.py:92-98
    rows = prod.reshape(1, -1).repeat(n, 1)                     # run1.py:112
    branch = nets["model4"](rows)                               # run1.py:113
    trunk = nets["model5"](stack_xt(x, t))                      # run1.py:114-115
    return torch.sum(branch * trunk, dim=1)                     # run1.py:116


def residual_net(nets, u1, u2, x, t, sigma: float, rate: float):
    """run1.py:120-129, derivatives taken the way the script takes them."""
    f = lambda xx, tt: operator_net(nets, u1, u2, xx, tt)
    s = f(x, t)
    s_x = jacrev(f, argnums=0)(x, t).sum(dim=0)
    s_xx = hessian(f, argnums=0)(x, t).sum(dim=0).sum(dim=0)
    s_t = jacrev(f, argnums=1)(x, t).sum(dim=0)
    return s_t - 0.5 * sigma ** 2 * x ** 2 * s_xx - rate * x * s_x + rate * s


def losses(nets, u1, u2, bc: Sequence[torch.Tensor], pde: Sequence[torch.Tensor], sigma: float, rate: float,
           bc_weight: float):
    """loss_bcs (run1.py:134-141), loss_res (:145-149) and their weighted sum (:224)."""
    x_i, t_i, out_i = bc
    x_b, t_b, out_b = pde
    s_pred = operator_net(nets, u1, u2, x_i, t_i)
    bc_loss = torch.mean((out_i.flatten() - s_pred) ** 2)
    res = residual_net(nets, u1, u2, x_b, t_b, sigma, rate)
    pde_loss = torch.mean((out_b.flatten() - res) ** 2)
    return {"s_pred": s_pred, "res": res, "bc_loss": bc_loss, "pde_loss": pde_loss,
            "loss": pde_loss + bc_weight * bc_loss}


def nets_from_layers(layer_sets: Dict[str, List[K.LayerParams]], act: str) -> Dict[str, Net]:
    return {name: (lambda z, L=layers: K.model_forward(z, L, act)) for name, layers in layer_sets.items()}


def step_with_grads(layer_sets: Dict[str, List[K.LayerParams]], act: str, u1, u2, bc, pde, sigma: float, rate: float,
                    bc_weight: float):
    """Losses plus d(loss)/d(every parameter), keyed ``model.layers.l.name``."""
    live, leaves, names = {}, [], []
    for mname, layers in layer_sets.items():
        copies = []
        for l, p in enumerate(layers):
            q = dict(p)
            for key in ("base_weight", "spline_weight", "spline_scaler"):
                if q.get(key) is not None:
                    q[key] = q[key].detach().clone().requires_grad_(True)
                    leaves.append(q[key])
                    names.append(f"{mname}.layers.{l}.{key}")
            copies.append(q)
        live[mname] = copies
    out = losses(nets_from_layers(live, act), u1, u2, bc, pde, sigma, rate, bc_weight)
    grads = torch.autograd.grad(out["loss"], leaves)
    res = {k: v.detach() for k, v in out.items()}
    res["grads"] = {n: g.detach() for n, g in zip(names, grads)}
    return res
