"""Physics-informed Black-Scholes training step on a :class:`pde_amd.KAN`.

What the reference scripts do per minibatch for the PDE part - ``residual_net``
(run1.py:120-129), ``loss_res`` (:145-149) and ``loss.backward()`` (:225) -
as one call into libpdekan.so: forward with analytic input derivatives,
residual, mean-squared loss, every parameter gradient.  The optimiser update
stays with ``torch.optim`` (the scripts use AdamW / LBFGS).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch

from . import _lib, functional as K
from ._lib import PdekanLayer, PdekanLayerGrad
from .kan import KAN, _act_kind

# run1.py:126-127
BS_SIGMA = 0.165856529
BS_RATE = 0.025610


class BlackScholesStep:
    """Reusable fused step for one model.

    Gradients land in ONE flat fp32 buffer (``flat_grad``) laid out in
    ``model.parameters()`` order; each parameter's ``.grad`` is a view into it,
    so an optimiser sees ordinary gradients and a data-parallel wrapper can
    all-reduce the whole model with a single collective.
    """

    def __init__(self, model: KAN, sigma: float = BS_SIGMA, rate: float = BS_RATE, micro_batch: int = 0,
                 workspace_budget: Optional[int] = None):
        """``micro_batch``: points per pass of the library over the batch (0 = all at once); the gradients of the
        passes are accumulated, the result is that of the whole batch up to summation order.  ``workspace_budget``
        (bytes): instead of a fixed ``micro_batch``, take the largest one (a multiple of 4096 points, halved until
        it fits) whose workspace stays within the budget; chosen per batch size at the first call."""
        self.model = model
        self.sigma = float(sigma)
        self.rate = float(rate)
        self.micro_batch = int(micro_batch)
        self.workspace_budget = None if workspace_budget is None else int(workspace_budget)
        if self.workspace_budget is not None and self.micro_batch:
            raise ValueError("give micro_batch or workspace_budget, not both")
        params = list(model.parameters())
        if not params or not params[0].is_cuda:
            raise RuntimeError("pde_amd: move the model to a ROCm GPU before building a BlackScholesStep "
                               "(there is no CPU implementation of the hot path)")
        self.device = params[0].device
        for layer in model.layers:
            if _act_kind(layer.base_activation) is None:
                raise NotImplementedError("the fused step supports SiLU and Identity base activations")
            if layer.spline_order != 3:
                raise NotImplementedError("the kernels implement cubic splines (spline_order=3)")
        if model.layers[-1].out_features != 1:
            raise ValueError("the PINN head must have one output")
        total = sum(p.numel() for p in params)
        # gradients and the loss share one buffer so that a data-parallel wrapper can all-reduce both in place
        self.grad_and_loss = torch.zeros(total + 1, dtype=torch.float32, device=self.device)
        self.flat_grad = self.grad_and_loss[:total]
        self._views: List[torch.Tensor] = []
        off = 0
        for p in params:
            v = self.flat_grad[off: off + p.numel()].view_as(p)
            off += p.numel()
            p.grad = v
            self._views.append(v)
        self.loss = self.grad_and_loss[total]              # 0-d view of the last element
        self._ws: Optional[torch.Tensor] = None
        self._ws_key = None

    # -- C structs are rebuilt per call: parameters may have been re-allocated
    def _structs(self):
        n = len(self.model.layers)
        layers = (PdekanLayer * n)()
        grads = (PdekanLayerGrad * n)()
        by_id = {id(p): v for p, v in zip(self.model.parameters(), self._views)}
        for i, layer in enumerate(self.model.layers):
            grid, bw, sw, sc = layer._kernel_params()
            for t in (grid, bw, sw) + ((sc,) if sc is not None else ()):
                if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                    raise RuntimeError("pde_amd: parameters must be contiguous float32 tensors on the GPU")
            layers[i] = K.layer_struct(grid, bw, sw, sc, _act_kind(layer.base_activation), layer.grid_size,
                                       layer.spline_order)
            grads[i] = PdekanLayerGrad(by_id[id(bw)].data_ptr(), by_id[id(sw)].data_ptr(),
                                       0 if sc is None else by_id[id(sc)].data_ptr())
        return layers, grads, n

    def _fit_micro_batch(self, lib, layers, n_layers, n, d) -> int:
        """Largest micro-batch (0 = the whole batch) whose workspace fits the budget."""
        mb = 0
        while True:
            need = lib.pdekan_model_workspace_bytes(layers, n_layers, n, d, d - 1, mb)
            if need == 0:
                raise _lib.PdekanError(lib.pdekan_last_error().decode())
            if need <= self.workspace_budget:
                return mb
            cur = n if mb == 0 else mb
            if cur <= 4096:
                raise _lib.PdekanError(f"pde_amd: a workspace budget of {self.workspace_budget} bytes does not hold "
                                       f"4096 points of this model ({need} bytes)")
            mb = max(4096, (cur // 2 + 4095) // 4096 * 4096)

    def _workspace(self, lib, layers, n_layers, n, d):
        if self.workspace_budget is not None and (self._ws_key is None or self._ws_key[:2] != (n, d)):
            self.micro_batch = self._fit_micro_batch(lib, layers, n_layers, n, d)
        key = (n, d, self.micro_batch)
        if self._ws is None or self._ws_key != key:
            need = lib.pdekan_model_workspace_bytes(layers, n_layers, n, d, d - 1, self.micro_batch)
            if need == 0:
                raise _lib.PdekanError(lib.pdekan_last_error().decode())
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws_key = key
        return self._ws

    def __call__(self, x: torch.Tensor, target: Optional[torch.Tensor] = None, n_global: Optional[int] = None,
                 residual_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Run the step on points ``x [N, d]`` (asset columns first, time last).

        Returns the loss (a 0-d device tensor, no host sync); parameter
        gradients are in ``p.grad`` / ``flat_grad``.  ``n_global`` is the
        divisor of the mean when ``x`` is one shard of a larger batch.
        """
        lib = _lib.load()
        K._require_gpu(x, "x")
        assert x.dim() == 2 and x.size(1) == self.model.layers[0].in_features
        n, d = x.shape
        if target is not None:
            target = target.detach().reshape(-1).contiguous()
            K._require_gpu(target, "target")
            assert target.numel() == n
        if residual_out is not None:
            assert residual_out.is_cuda and residual_out.is_contiguous() and residual_out.numel() == n
        for p, v in zip(self.model.parameters(), self._views):
            if p.grad is not v:            # e.g. after optimizer.zero_grad(set_to_none=True)
                p.grad = v
        layers, grads, nl = self._structs()
        xd = x.detach()
        with torch.cuda.device(self.device):
            ws = self._workspace(lib, layers, nl, n, d)
            _lib.check(lib.pdekan_bs_step(layers, nl, K._ptr(xd), xd.stride(0), xd.stride(1), K._ptr(target), n,
                                          n if n_global is None else int(n_global), self.sigma, self.rate,
                                          K._ptr(self.loss), K._ptr(residual_out), grads, 0, self.micro_batch,
                                          K._ptr(ws), ws.numel(), K._stream(self.device)))
        return self.loss


class GraphedStep:
    """The fused step on ONE fixed batch, captured in a HIP graph and replayed per call.

    For optimisers that evaluate the same batch many times per update - the LBFGS scripts call their closure up to
    25 times per step (run2.py:174-199, kan/LBFGS.py:228-229) - a replay costs one graph launch instead of a
    dozen kernel launches plus the host code around them.  Parameters are read from their own storage at replay,
    so in-place updates (what torch.optim does) are seen; the batch is the captured copy ``x`` (``copy_`` into it to
    change points without re-capturing).  Use as an LBFGS closure::

        g = GraphedStep(BlackScholesStep(model), x)
        def closure():
            opt.zero_grad()
            return g()
    """

    def __init__(self, step: BlackScholesStep, x: torch.Tensor, target: Optional[torch.Tensor] = None,
                 n_global: Optional[int] = None):
        self.step = step
        self.x = x.detach().clone()
        self.target = None if target is None else target.detach().clone()
        self.n_global = n_global
        step(self.x, target=self.target, n_global=n_global)        # allocates the workspace outside the capture
        torch.cuda.synchronize(step.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            step(self.x, target=self.target, n_global=n_global)

    def __call__(self) -> torch.Tensor:
        for p, v in zip(self.step.model.parameters(), self.step._views):
            if p.grad is not v:            # e.g. after optimizer.zero_grad(set_to_none=True)
                p.grad = v
        self.graph.replay()
        return self.step.loss


def bs_residual_from_jet(x: torch.Tensor, jet: torch.Tensor, sigma: float = BS_SIGMA, rate: float = BS_RATE):
    """Residual of run1.py:128 from ``KAN.forward_jet`` output ``[N, 1, J]``
    in ordinary tensor ops (a handful of [N]-sized ops; differentiable)."""
    d = x.size(1)
    v = jet[:, 0, :]
    res = v[:, d] + rate * v[:, 0]
    for j in range(d - 1):
        res = res - 0.5 * sigma ** 2 * x[:, j] ** 2 * v[:, 1 + d + j] - rate * x[:, j] * v[:, 1 + j]
    return res
