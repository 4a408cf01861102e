"""``KANLinear`` / ``KAN`` with the reference's module interface, running on
hand-written gfx950 kernels.

Interface mirrored (constructor arguments, attribute / parameter / buffer
names, ``state_dict`` keys, method names, argument meaning, assertion
behaviour): efficient_kan/kan.py:6-237 (``KANLinear``) and :240-285 (``KAN``).
What differs is underneath: ``forward`` and its backward are HIP kernels behind
the C ABI of ``libpdekan.so`` (see ``include/pdekan.h``); there is no PyTorch
or CPU implementation of the hot path here, so a forward on a non-GPU tensor
raises.  Two additions that the reference does not have:

* ``forward_jet`` carries analytic input derivatives (what the scripts obtain
  with ``jacrev`` / ``hessian``, run1.py:121-124) through the stack in one pass;
* ``pde_amd.pinn.BlackScholesStep`` fuses forward, PDE residual, loss and
  all parameter gradients into one library call.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import functional as K
from .spline_host import dense_bsplines, extended_knots, fit_coefficients

__all__ = ["KANLinear", "KAN"]


def _under_func_transform(t: torch.Tensor) -> bool:
    """True while ``t`` is a torch.func wrapper (grad / vmap / jvp level)."""
    fn = getattr(torch._C._functorch, "is_functorch_wrapped_tensor", None)
    return bool(fn and fn(t))


def _act_kind(module: torch.nn.Module) -> Optional[int]:
    """Kernel id of a base activation, or None when the kernels do not know it."""
    if type(module) is torch.nn.Identity:
        return K.ACT_IDENTITY
    if type(module) is torch.nn.SiLU:
        return K.ACT_SILU
    return None


class KANLinear(torch.nn.Module):
    """One KAN layer: ``y = W_b act(x) + (W_s * scaler) B(x)`` (kan.py:6-54)."""

    def __init__(self, in_features, out_features, grid_size=5, spline_order=3, scale_noise=0.1,
                 scale_base=1.0, scale_spline=1.0, enable_standalone_scale_spline=True,
                 base_activation=torch.nn.SiLU, grid_eps=0.02, grid_range=[-1, 1]):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.grid_size = grid_size
        self.spline_order = spline_order

        self.register_buffer("grid", extended_knots(in_features, grid_size, spline_order, grid_range))
        self.base_weight = torch.nn.Parameter(torch.empty(out_features, in_features))
        self.spline_weight = torch.nn.Parameter(
            torch.empty(out_features, in_features, grid_size + spline_order))
        if enable_standalone_scale_spline:
            self.spline_scaler = torch.nn.Parameter(torch.empty(out_features, in_features))

        self.scale_noise = scale_noise
        self.scale_base = scale_base
        self.scale_spline = scale_spline
        self.enable_standalone_scale_spline = enable_standalone_scale_spline
        self.base_activation = base_activation()
        self.grid_eps = grid_eps
        self.reset_parameters()

    # ---- initialisation: consumes the RNG in the reference's order (kan.py:56-76)
    def reset_parameters(self):
        torch.nn.init.kaiming_uniform_(self.base_weight, a=math.sqrt(5) * self.scale_base)
        with torch.no_grad():
            jitter = (torch.rand(self.grid_size + 1, self.in_features, self.out_features) - 1 / 2) \
                * self.scale_noise / self.grid_size
            interior = self.grid.T[self.spline_order: -self.spline_order]
            gain = 1.0 if self.enable_standalone_scale_spline else self.scale_spline
            self.spline_weight.data.copy_(gain * self.curve2coeff(interior, jitter))
            if self.enable_standalone_scale_spline:
                torch.nn.init.kaiming_uniform_(self.spline_scaler, a=math.sqrt(5) * self.scale_spline)

    # ---- non-hot methods, plain tensor code
    def b_splines(self, x: torch.Tensor):
        """Dense bases ``(batch, in_features, grid_size + spline_order)`` (kan.py:78-111)."""
        assert x.dim() == 2 and x.size(1) == self.in_features
        if x.is_cuda and 0 <= self.spline_order <= 5 and x.dtype == torch.float32:
            return K.basis_dense(x, self.grid, self.grid_size, self.spline_order)     # the library's basis kernel
        return dense_bsplines(x, self.grid, self.spline_order)

    def curve2coeff(self, x: torch.Tensor, y: torch.Tensor):
        """Coefficients interpolating samples (kan.py:113-143)."""
        assert x.dim() == 2 and x.size(1) == self.in_features
        assert y.size() == (x.size(0), self.in_features, self.out_features)
        if K.fit_on_device(x, self.grid_size, self.spline_order) and y.dtype == torch.float32:
            coeff = K.fit_coefficients(x, self.grid, self.grid_size, self.spline_order, y=y)   # fp64 solve in one kernel
        else:
            coeff = fit_coefficients(x, y, self.grid, self.spline_order)
        assert coeff.size() == (self.out_features, self.in_features, self.grid_size + self.spline_order)
        return coeff

    @property
    def scaled_spline_weight(self):
        """kan.py:145-151."""
        if self.enable_standalone_scale_spline:
            return self.spline_weight * self.spline_scaler.unsqueeze(-1)
        return self.spline_weight * 1.0

    # ---- hot path
    def _kernel_params(self):
        scaler = self.spline_scaler if self.enable_standalone_scale_spline else None
        return self.grid, self.base_weight, self.spline_weight, scaler

    def forward(self, x: torch.Tensor):
        """kan.py:153-166: any leading dimensions, last one = ``in_features``."""
        assert x.size(-1) == self.in_features
        lead = x.shape[:-1]
        flat = x.reshape(-1, self.in_features)
        kind = _act_kind(self.base_activation)
        grid, bw, sw, sc = self._kernel_params()
        if _under_func_transform(flat) or self.spline_order != 3:
            # Spline orders other than 3 (no reference caller uses one; the whole-layer kernels are cubic) and
            # torch.func transforms (the scripts' jacrev / hessian, run1.py:122-124) trace through tensor ops and
            # cannot see inside a whole-layer library call.  Under them the layer is its defining expression
            # (kan.py:153-166): two GEMMs around the dense bases.  On a GPU the bases come from the library's
            # basis kernel, an elementwise op that carries its own derivative, vmap and forward-mode rules
            # (functional.basis_dense), so jacrev / hessian of any depth stay on it; on a CPU tensor or in another
            # dtype than float32 (no kernels there) they are plain tensor code.  O(n^2)/O(n^3) like the scripts either way; forward_jet gives the
            # same derivatives in O(n).
            if flat.is_cuda and grid.is_cuda and flat.dtype == torch.float32 and 0 <= self.spline_order <= 5:
                bases = K.basis_dense(flat, grid, self.grid_size, self.spline_order)
            else:
                bases = dense_bsplines(flat, grid, self.spline_order)
            base = F.linear(self.base_activation(flat), bw)
            spline = F.linear(bases.reshape(flat.size(0), -1), self.scaled_spline_weight.view(self.out_features, -1))
            return (base + spline).view(*lead, self.out_features)
        if kind is not None:
            out = K.kan_layer(flat, grid, bw, sw, sc, kind, self.grid_size, self.spline_order)
        else:
            # an activation the kernels do not implement: the spline branch still
            # runs on the kernels (base weight zeroed), the base branch is two ATen ops
            out = K.kan_layer(flat, grid, torch.zeros_like(bw), sw, sc, K.ACT_IDENTITY,
                              self.grid_size, self.spline_order)
            out = out + F.linear(self.base_activation(flat), bw)
        return out.view(*lead, self.out_features)

    def forward_jet(self, x_jet: torch.Tensor, n_dir: int, n_second: int):
        """Layer forward on jets ``[N, in_features, 1+n_dir+n_second]``."""
        kind = _act_kind(self.base_activation)
        if kind is None:
            raise NotImplementedError("forward_jet supports SiLU and Identity base activations")
        grid, bw, sw, sc = self._kernel_params()
        return K.kan_layer_jet(x_jet, grid, bw, sw, sc, kind, self.grid_size, self.spline_order,
                               n_dir, n_second)

    # ---- grid adaptation (kan.py:168-215); rarely called, ordinary tensor code
    @torch.no_grad()
    def update_grid(self, x: torch.Tensor, margin=0.01):
        assert x.dim() == 2 and x.size(1) == self.in_features
        batch = x.size(0)
        k, g = self.spline_order, self.grid_size

        on_device = K.fit_on_device(x, g, k) and self.grid.is_cuda
        if on_device:
            # the refit kernel evaluates the edge functions on the old grid itself: keep the old grid, skip the
            # (batch, in, out) tensor
            old_grid = self.grid.clone()
        else:
            # per-edge spline outputs at the samples, (batch, in, out)
            bases = self.b_splines(x).permute(1, 0, 2)
            coeff = self.scaled_spline_weight.permute(1, 2, 0)
            edge_out = torch.bmm(bases, coeff).permute(1, 0, 2)

        ordered = torch.sort(x, dim=0)[0]
        picks = torch.linspace(0, batch - 1, g + 1, dtype=torch.int64, device=x.device)
        adaptive = ordered[picks]
        step = (ordered[-1] - ordered[0] + 2 * margin) / g
        uniform = torch.arange(g + 1, dtype=torch.float32, device=x.device).unsqueeze(1) * step \
            + ordered[0] - margin
        inner = self.grid_eps * uniform + (1 - self.grid_eps) * adaptive
        below = inner[:1] - step * torch.arange(k, 0, -1, device=x.device).unsqueeze(1)
        above = inner[-1:] + step * torch.arange(1, k + 1, device=x.device).unsqueeze(1)
        knots = torch.concatenate([below, inner, above], dim=0)

        self.grid.copy_(knots.T)
        if on_device:
            scaler = self.spline_scaler if self.enable_standalone_scale_spline else None
            self.spline_weight.data.copy_(K.fit_coefficients(x, self.grid, g, k, old_grid=old_grid,
                                                             old_spline_weight=self.spline_weight,
                                                             old_spline_scaler=scaler))
        else:
            self.spline_weight.data.copy_(self.curve2coeff(x, edge_out))

    def regularization_loss(self, regularize_activation=1.0, regularize_entropy=1.0):
        """Weight-based stand-in for the paper's activation L1 + entropy terms
        (kan.py:217-237)."""
        edge_l1 = self.spline_weight.abs().mean(-1)
        total = edge_l1.sum()
        share = edge_l1 / total
        entropy = -torch.sum(share * share.log())
        return regularize_activation * total + regularize_entropy * entropy


class KAN(torch.nn.Module):
    """Sequential stack of :class:`KANLinear` (kan.py:240-285)."""

    def __init__(self, layers_hidden, grid_size=5, spline_order=3, scale_noise=0.1, scale_base=1.0,
                 scale_spline=1.0, base_activation=torch.nn.SiLU, grid_eps=0.02, grid_range=[-1, 1]):
        super().__init__()
        self.grid_size = grid_size
        self.spline_order = spline_order
        self.layers = torch.nn.ModuleList(
            KANLinear(n_in, n_out, grid_size=grid_size, spline_order=spline_order, scale_noise=scale_noise,
                      scale_base=scale_base, scale_spline=scale_spline, base_activation=base_activation,
                      grid_eps=grid_eps, grid_range=grid_range)
            for n_in, n_out in zip(layers_hidden, layers_hidden[1:]))

    def forward(self, x: torch.Tensor, update_grid=False):
        for layer in self.layers:
            if update_grid:
                layer.update_grid(x)
            x = layer(x)
        return x

    def forward_jet(self, x: torch.Tensor, n_second: Optional[int] = None):
        """Values and analytic input derivatives in one pass.

        ``x`` is ``[N, d]`` (any strides).  Returns ``[N, out, 1 + d + n_second]``:
        slot 0 the output, slots ``1..d`` its derivative with respect to each
        input column, the rest the second derivative with respect to the first
        ``n_second`` columns (default ``d - 1``: all but the last, which the
        PINN scripts use as time).  Differentiable with respect to the
        parameters.
        """
        assert x.dim() == 2 and x.size(1) == self.layers[0].in_features
        d = x.size(1)
        if n_second is None:
            n_second = d - 1
        jet = K.seed_jets(x, d, n_second)
        for layer in self.layers:
            jet = layer.forward_jet(jet, d, n_second)
        return jet

    def regularization_loss(self, regularize_activation=1.0, regularize_entropy=1.0):
        return sum(layer.regularization_loss(regularize_activation, regularize_entropy)
                   for layer in self.layers)
