"""CPU oracle for the KAN-PINN hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU with plain PyTorch ops, the algorithm of the
reference's ``efficient_kan`` layer and of the Black-Scholes residual the
reference scripts build around it.  It exists to CHECK the HIP kernels; nothing
in the product package imports it.  Only ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py`` may call into it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
module itself (``/root/reference/efficient_kan``) in the build container and
stores its outputs as fixtures under ``tests/golden/``;
``tests/test_oracle_golden.py`` holds this oracle to those fixtures.

Reference lines followed (paths relative to the reference checkout):
  * dense Cox-de Boor basis ............ efficient_kan/kan.py:78-111
  * scaled spline weight ............... efficient_kan/kan.py:145-151
  * layer forward ...................... efficient_kan/kan.py:153-166
  * model forward ...................... efficient_kan/kan.py:274-279
  * extended uniform knot vector ....... efficient_kan/kan.py:27-36
  * Black-Scholes residual ............. run1.py:120-129 (also run2.py:131-140)
  * mean-squared residual loss ......... run1.py:145-149
The scripts take s_x / s_t / s_xx with ``jacrev``/``hessian`` of a batched
function and then sum away the off-diagonal zeros (run1.py:122-124).  Points are
independent, so the same numbers come from differentiating ``sum(V)`` with
respect to the point coordinates; that is what :func:`point_jets` does.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

LayerParams = Dict[str, torch.Tensor]

# constants of run1.py:126-127
BS_SIGMA = 0.165856529
BS_RATE = 0.025610


# --------------------------------------------------------------------------
# model pieces
# --------------------------------------------------------------------------
def uniform_grid(in_features: int, grid_size: int, spline_order: int = 3,
                 grid_range=(-1.0, 1.0), dtype=torch.float32) -> torch.Tensor:
    """Extended uniform knots, one row per input feature (kan.py:27-36)."""
    h = (grid_range[1] - grid_range[0]) / grid_size
    knots = torch.arange(-spline_order, grid_size + spline_order + 1) * h + grid_range[0]
    return knots.expand(in_features, -1).contiguous().to(dtype)


def b_splines(x: torch.Tensor, grid: torch.Tensor, spline_order: int) -> torch.Tensor:
    """All G+k order-k B-spline bases of every feature, dense (kan.py:78-111).

    x [N, I], grid [I, G+2k+1]  ->  [N, I, G+k].  Order-0 bases are indicators
    of the half-open knot intervals; a point outside the extended grid gets all
    zeros.  Each level divides by knot differences taken from ``grid`` itself.
    """
    assert x.dim() == 2 and x.size(1) == grid.size(0)
    xe = x.unsqueeze(-1)
    b = ((xe >= grid[:, :-1]) & (xe < grid[:, 1:])).to(x.dtype)
    for p in range(1, spline_order + 1):
        lo = (xe - grid[:, : -(p + 1)]) / (grid[:, p:-1] - grid[:, : -(p + 1)]) * b[:, :, :-1]
        hi = (grid[:, p + 1:] - xe) / (grid[:, p + 1:] - grid[:, 1:(-p)]) * b[:, :, 1:]
        b = lo + hi
    return b.contiguous()


def activation(name: str):
    if name == "silu":
        return F.silu
    if name == "identity":
        return lambda t: t
    raise ValueError(f"oracle knows silu/identity only, got {name!r}")


def layer_forward(x: torch.Tensor, p: LayerParams, act: str, spline_order: int = 3) -> torch.Tensor:
    """One KANLinear forward on a 2-D batch (kan.py:153-166)."""
    out_f, in_f = p["base_weight"].shape
    w = p["spline_weight"]
    if p.get("spline_scaler") is not None:
        w = w * p["spline_scaler"].unsqueeze(-1)          # kan.py:145-151
    base = F.linear(activation(act)(x), p["base_weight"])
    spl = F.linear(b_splines(x, p["grid"], spline_order).view(x.size(0), -1), w.view(out_f, -1))
    return base + spl


def model_forward(x: torch.Tensor, layers: Sequence[LayerParams], act: str,
                  spline_order: int = 3) -> torch.Tensor:
    """Sequential stack (kan.py:274-279, update_grid=False)."""
    for p in layers:
        x = layer_forward(x, p, act, spline_order)
    return x


# --------------------------------------------------------------------------
# per-point derivatives and the PDE residual
# --------------------------------------------------------------------------
def point_jets(x: torch.Tensor, layers: Sequence[LayerParams], act: str, n_second: int,
               spline_order: int = 3, create_graph: bool = True):
    """V, dV/dx_j for every input j, d2V/dx_j^2 for j < n_second, per point.

    Returns (V [N], first [N, d], second [N, n_second]).  Equivalent to the
    reference's jacrev/hessian-then-sum (run1.py:121-124) because the output
    for one point depends on that point alone.
    """
    assert x.requires_grad
    v = model_forward(x, layers, act, spline_order)
    assert v.size(1) == 1, "the PINN head is scalar"
    v = v[:, 0]
    (g1,) = torch.autograd.grad(v.sum(), x, create_graph=True)
    seconds = []
    for j in range(n_second):
        (g2,) = torch.autograd.grad(g1[:, j].sum(), x, create_graph=create_graph,
                                    retain_graph=True)
        seconds.append(g2[:, j])
    second = torch.stack(seconds, dim=1) if seconds else x.new_zeros(x.size(0), 0)
    return v, g1, second


def bs_residual(x: torch.Tensor, v, first, second, sigma: float = BS_SIGMA, rate: float = BS_RATE):
    """Black-Scholes residual in the scripts' normalised coordinates.

    The last input column is time, the others are asset prices.  For one asset
    this is run1.py:128: ``s_t - 0.5*sigma^2*x^2*s_xx - r*x*s_x + r*s``; with
    several assets the price terms are summed over the assets (diagonal second
    derivatives only).
    """
    d = x.size(1)
    n_assets = d - 1
    assert second.size(1) == n_assets
    res = first[:, d - 1] + rate * v
    for j in range(n_assets):
        s = x[:, j]
        res = res - 0.5 * sigma ** 2 * s ** 2 * second[:, j] - rate * s * first[:, j]
    return res


def pinn_step(x: torch.Tensor, layers: List[LayerParams], act: str, target: torch.Tensor | None = None,
              sigma: float = BS_SIGMA, rate: float = BS_RATE, n_global: int | None = None,
              spline_order: int = 3):
    """Forward + residual + mean-squared loss + parameter gradients.

    ``layers`` hold leaf tensors; gradients are returned as a list of dicts
    with the keys ``base_weight``, ``spline_weight``, ``spline_scaler``.
    ``n_global`` lets a shard divide by the whole job's point count
    (loss = sum(shard res^2) / n_global), which is how data-parallel ranks
    split run1.py:148's mean.
    """
    dt = x.dtype
    leaves = []
    live = []
    for p in layers:
        q = dict(p)
        for key in ("base_weight", "spline_weight", "spline_scaler"):
            if q.get(key) is not None:
                q[key] = q[key].detach().clone().requires_grad_(True)
                leaves.append(q[key])
        live.append(q)
    xr = x.detach().clone().requires_grad_(True)
    n_second = x.size(1) - 1
    v, first, second = point_jets(xr, live, act, n_second, spline_order)
    res = bs_residual(xr, v, first, second, sigma, rate)
    tgt = torch.zeros_like(res) if target is None else target.to(dt).flatten()
    n = x.size(0) if n_global is None else n_global
    loss = ((tgt - res) ** 2).sum() / n
    grads = torch.autograd.grad(loss, leaves)
    out, it = [], iter(grads)
    for q in live:
        g = {}
        for key in ("base_weight", "spline_weight", "spline_scaler"):
            g[key] = next(it).detach() if q.get(key) is not None else None
        out.append(g)
    return {
        "v": v.detach(), "first": first.detach(), "second": second.detach(),
        "res": res.detach(), "loss": loss.detach(), "grads": out,
    }


def value_step(x: torch.Tensor, layers: List[LayerParams], act: str, gy: torch.Tensor,
               spline_order: int = 3):
    """Plain forward and first-order backward: y = model(x), cotangent gy.

    Gives d<gy, y>/d(params) and d<gy, y>/dx - what ``loss.backward()`` sees
    through ``KAN.forward`` in a data-fit loss such as run1.py:134-141.
    """
    leaves, live = [], []
    for p in layers:
        q = dict(p)
        for key in ("base_weight", "spline_weight", "spline_scaler"):
            if q.get(key) is not None:
                q[key] = q[key].detach().clone().requires_grad_(True)
                leaves.append(q[key])
        live.append(q)
    xr = x.detach().clone().requires_grad_(True)
    y = model_forward(xr, live, act, spline_order)
    grads = torch.autograd.grad((y * gy).sum(), leaves + [xr])
    out, it = [], iter(grads)
    for q in live:
        g = {}
        for key in ("base_weight", "spline_weight", "spline_scaler"):
            g[key] = next(it).detach() if q.get(key) is not None else None
        out.append(g)
    return {"y": y.detach(), "grads": out, "gx": grads[-1].detach()}


# --------------------------------------------------------------------------
# helpers shared by tests and the CPU baseline
# --------------------------------------------------------------------------
def layers_from_state_dict(sd: Dict[str, torch.Tensor], dtype=None) -> List[LayerParams]:
    """Group ``layers.{l}.{name}`` entries (the reference's state_dict layout)."""
    n = 1 + max(int(k.split(".")[1]) for k in sd)
    out = []
    for l in range(n):
        p = {}
        for name in ("grid", "base_weight", "spline_weight", "spline_scaler"):
            t = sd.get(f"layers.{l}.{name}")
            if t is not None and dtype is not None:
                t = t.to(dtype)
            p[name] = t
        out.append(p)
    return out


def random_layers(widths: Sequence[int], grid_size: int, seed: int, spline_order: int = 3,
                  dtype=torch.float32, scale: float = 0.3) -> List[LayerParams]:
    """Generic (not reference-initialised) parameters for throughput timing."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i, o in zip(widths, widths[1:]):
        c = grid_size + spline_order
        out.append({
            "grid": uniform_grid(i, grid_size, spline_order, dtype=dtype),
            "base_weight": ((torch.rand(o, i, generator=g) - 0.5) * 2 / math.sqrt(i)).to(dtype),
            "spline_weight": ((torch.rand(o, i, c, generator=g) - 0.5) * scale).to(dtype),
            "spline_scaler": ((torch.rand(o, i, generator=g) - 0.5) * 2 / math.sqrt(i)).to(dtype),
        })
    return out
