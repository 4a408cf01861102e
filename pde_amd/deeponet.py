"""Physics-informed DeepONet over :class:`pde_amd.KAN` networks.

Mirrors the ``PI_DeepONet`` class the reference scripts build around their four
KANs (run1.py:75-149; same in run2.py:86-160, run3.py:87-161): two sensor
networks whose outputs are multiplied (``brunk_net``), a mixing network on the
flattened product (``model4``), a trunk network on ``(x, t)`` (``model5``), the
prediction ``sum(branch * trunk, dim=1)``, the Black-Scholes residual and the
two mean-squared losses.  Method names, argument order and return shapes are
the reference's, so its training loop (run1.py:214-233) runs on this class
unchanged.

Two things are done differently, with identical results:

* the reference feeds ``model4`` n copies of the same row
  (``B1.view(1, -1).repeat(n, 1)``, run1.py:112); every row gives the same
  output, so ``model4`` is evaluated once on the single row and broadcast;
* the derivatives ``s_x, s_t, s_xx`` come from the trunk's input jets
  (``KAN.forward_jet``, one O(n) kernel pass) instead of ``jacrev`` / ``hessian``
  over the whole batch (run1.py:122-124, O(n^2) / O(n^3) temporaries).  The
  branch factor does not depend on ``(x, t)``, so the jet of the prediction is the
  branch factor times the jet of the trunk.

``train_step`` (the loop body run1.py:217-233) runs as ONE library call of four launches
(:class:`DeepONetStep`, ``pdekan_deeponet_step``) when the four networks are small enough for it - the scripts'
own are (run1.py:478-482) - and otherwise layer by layer under autograd like the methods above.

Everything runs on the library kernels: there is no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch
from torch import nn

from . import _lib, functional as K
from ._lib import PdekanLayer, PdekanLayerGrad, PdekanNet
from .kan import KAN, _act_kind
from .pinn import BS_RATE, BS_SIGMA


class DeepONetStep:
    """Losses and every parameter gradient of one minibatch of a :class:`PI_DeepONet` in one library call.

    Gradients land in one flat fp32 buffer laid out in ``net.parameters()`` order; each parameter's ``.grad`` is a
    view into it (overwritten by every call).  ``losses`` is a 3-element device tensor: total, PDE, boundary.
    """

    def __init__(self, net: "PI_DeepONet"):
        self.net = net
        self.models = [net.model1, net.model2, net.model4, net.model5]
        params = list(net.parameters())
        if not params or not params[0].is_cuda:
            raise RuntimeError("pde_amd: move the networks to a ROCm GPU first (there is no CPU implementation)")
        self.device = params[0].device
        self.flat_grad = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=self.device)
        self.losses = torch.zeros(3, dtype=torch.float32, device=self.device)
        self._params = params
        self._views: List[torch.Tensor] = []
        off = 0
        for p in params:
            self._views.append(self.flat_grad[off: off + p.numel()].view_as(p))
            off += p.numel()
        self._ws = {}
        self._keep = None
        self.calls = 0                         # steps the library call took

    def supported(self) -> bool:
        """Whether the networks are of a kind the call takes at all (sizes are settled per batch)."""
        for m in self.models:
            for layer in m.layers:
                if _act_kind(layer.base_activation) is None or layer.spline_order != 3:
                    return False
        return all(p.requires_grad and p.dtype == torch.float32 for p in self._params)

    def _structs(self):
        by_id = {id(p): v for p, v in zip(self._params, self._views)}
        nets = (PdekanNet * 4)()
        keep = []
        for k, model in enumerate(self.models):
            n = len(model.layers)
            layers = (PdekanLayer * n)()
            grads = (PdekanLayerGrad * n)()
            for i, layer in enumerate(model.layers):
                grid, bw, sw, sc = layer._kernel_params()
                for t in (grid, bw, sw) + ((sc,) if sc is not None else ()):
                    if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                        raise RuntimeError("pde_amd: parameters must be contiguous float32 tensors on the GPU")
                layers[i] = K.layer_struct(grid, bw, sw, sc, _act_kind(layer.base_activation), layer.grid_size,
                                           layer.spline_order)
                grads[i] = PdekanLayerGrad(by_id[id(bw)].data_ptr(), by_id[id(sw)].data_ptr(),
                                           0 if sc is None else by_id[id(sc)].data_ptr())
            nets[k] = PdekanNet(layers, grads, n, 0)
            keep += [layers, grads]
        self._keep = keep                      # the arrays nets points into
        return nets

    def workspace(self, nets, m: int, n_bc: int, n_pde: int) -> Optional[torch.Tensor]:
        """The call's workspace for these sizes, or None when the library refuses them."""
        key = (m, n_bc, n_pde)
        if key not in self._ws:
            need = _lib.load().pdekan_deeponet_workspace_bytes(nets, m, n_bc, n_pde)
            self._ws[key] = torch.empty(need, dtype=torch.uint8, device=self.device) if need else None
        return self._ws[key]

    def __call__(self, u1, u2, bc_batch, pde_batch, s_pred=None, res_out=None) -> Optional[torch.Tensor]:
        """Returns ``losses`` (no host sync), or None when these networks / sizes have to go layer by layer."""
        lib = _lib.load()
        f = lambda t, what: K._prep(t.reshape(-1), what)
        u1, u2 = K._prep(u1, "u1"), K._prep(u2, "u2")
        x_i, t_i, o_i = (f(t, "boundary batch") for t in bc_batch)
        x_b, t_b, o_b = (f(t, "collocation batch") for t in pde_batch)
        m, n_bc, n_pde = u1.size(0), x_i.numel(), x_b.numel()
        if u1.dim() != 2 or u2.dim() != 2 or u2.size(0) != m or not (t_i.numel() == o_i.numel() == n_bc) \
                or not (t_b.numel() == o_b.numel() == n_pde):
            return None
        if u1.size(1) != self.net.model1.layers[0].in_features or u2.size(1) != self.net.model2.layers[0].in_features:
            return None
        nets = self._structs()
        with torch.cuda.device(self.device):
            ws = self.workspace(nets, m, n_bc, n_pde)
            if ws is None:
                return None
            for p, v in zip(self._params, self._views):
                if p.grad is not v:
                    p.grad = v
            _lib.check(lib.pdekan_deeponet_step(nets, K._ptr(u1), K._ptr(u2), m, K._ptr(x_i), K._ptr(t_i), K._ptr(o_i),
                                                n_bc, K._ptr(x_b), K._ptr(t_b), K._ptr(o_b), n_pde, self.net.sigma,
                                                self.net.rate, self.net.bc_weight, K._ptr(self.losses), K._ptr(s_pred),
                                                K._ptr(res_out), K._ptr(ws), ws.numel(), K._stream(self.device)))
        self.calls += 1
        return self.losses


class PI_DeepONet(nn.Module):
    def __init__(self, model1: KAN, model2: KAN, model4: KAN, model5: KAN, sigma: float = BS_SIGMA,
                 rate: float = BS_RATE, bc_weight: float = 8.0):
        super().__init__()
        self.model1 = model1
        self.model2 = model2
        self.model4 = model4
        self.model5 = model5
        self.sigma = float(sigma)          # run1.py:126
        self.rate = float(rate)            # run1.py:127
        self.bc_weight = float(bc_weight)  # run1.py:224
        self.bc_losses = []
        self.pde_losses = []
        self.fused = True                  # train_step as one library call where the networks allow it
        self._step: Optional[DeepONetStep] = None
        if model4.layers[-1].out_features != model5.layers[-1].out_features:
            raise ValueError("model4 and model5 must have the same output width (their product is summed over it)")
        if model5.layers[0].in_features != 2:
            raise ValueError("the trunk network takes (x, t)")

    # ---- pieces with the reference's names ---------------------------------
    def reshape(self, X: torch.Tensor) -> torch.Tensor:
        return X.reshape(-1,)

    def brunk_net(self, u1, u2, u_s1=None, u_s2=None) -> torch.Tensor:
        """run1.py:92-98: product of the two sensor networks, [sensors, 1]."""
        return self.model1(u1) * self.model2(u2)

    def helper(self, X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
        """run1.py:101-106: [n, 2] view with strides (1, n); the kernels read it in place."""
        return torch.stack([self.reshape(X), self.reshape(Y)]).permute(1, 0)

    def branch_row(self, u1, u2, u_s1=None, u_s2=None) -> torch.Tensor:
        """model4 on the flattened sensor product, evaluated once: [1, p]."""
        prod = self.brunk_net(u1, u2, u_s1, u_s2)
        return self.model4(prod.reshape(1, -1).float())

    # the three evaluations below take the branch row B [1, p] as an argument so that one
    # training step can share it between the boundary and the residual loss
    def _predict(self, B, x, t) -> torch.Tensor:
        T = self.model5(self.helper(x, t))
        return torch.sum(B * T, dim=1)

    def _jet(self, B, x, t) -> torch.Tensor:
        T = self.model5.forward_jet(self.helper(x, t))                # [n, p, 4]
        return torch.sum(B.unsqueeze(-1) * T, dim=1)

    def _residual(self, B, x, t) -> torch.Tensor:
        jet = self._jet(B, x, t)
        s, s_x, s_t, s_xx = jet[:, 0], jet[:, 1], jet[:, 2], jet[:, 3]
        x = self.reshape(x)
        return s_t - 0.5 * self.sigma ** 2 * x ** 2 * s_xx - self.rate * x * s_x + self.rate * s

    def operator_net(self, u1, u2, u_s1, u_s2, x, t) -> torch.Tensor:
        """run1.py:109-117: prediction at the n points, shape [n]."""
        return self._predict(self.branch_row(u1, u2, u_s1, u_s2), x, t)

    def operator_jet(self, u1, u2, u_s1, u_s2, x, t) -> torch.Tensor:
        """(s, s_x, s_t, s_xx) at the n points, shape [n, 4]."""
        return self._jet(self.branch_row(u1, u2, u_s1, u_s2), x, t)

    def residual_net(self, u1, u2, u_s1, u_s2, x, t) -> torch.Tensor:
        """run1.py:120-129."""
        return self._residual(self.branch_row(u1, u2, u_s1, u_s2), x, t)

    def loss_bcs(self, u1, u2, u_s1, u_s2, x, t, output) -> torch.Tensor:
        """run1.py:134-141."""
        s_pred = self.operator_net(u1, u2, u_s1, u_s2, x, t)
        return torch.mean((output.flatten() - s_pred) ** 2)

    def loss_res(self, u1, u2, u_s1, u_s2, x, t, output) -> torch.Tensor:
        """run1.py:145-149."""
        pred = self.residual_net(u1, u2, u_s1, u_s2, x, t)
        return torch.mean((output.flatten() - pred) ** 2)

    # ---- one optimiser step, as the loop body of run1.py:217-233 -----------
    def train_step(self, optimizer: torch.optim.Optimizer, u1, u2, u_s1, u_s2, bc_batch: Sequence[torch.Tensor],
                   pde_batch: Sequence[torch.Tensor]):
        if self.fused:
            if self._step is None or self._step._params[0].device != next(self.parameters()).device:
                self._step = DeepONetStep(self)
            if self._step.supported():
                losses = self._step(u1, u2, bc_batch, pde_batch)
                if losses is not None:
                    optimizer.step()
                    out = losses.clone()       # (the buffer is overwritten by the next step)
                    return out[0], out[1], out[2]
        x_i, t_i, outputs_i = bc_batch
        x_b, t_b, outputs_b = pde_batch
        optimizer.zero_grad()
        B = self.branch_row(u1, u2, u_s1, u_s2)          # the same for both losses: evaluated once per step
        bc_loss = torch.mean((outputs_i.flatten() - self._predict(B, x_i, t_i)) ** 2)
        pde_loss = torch.mean((outputs_b.flatten() - self._residual(B, x_b, t_b)) ** 2)
        loss = pde_loss + self.bc_weight * bc_loss
        loss.backward()
        optimizer.step()
        return loss.detach(), pde_loss.detach(), bc_loss.detach()

    def fit(self, u1, u2, u_s1, u_s2, dataloader1, dataloader2, epochs: int = 1500, lr: float = 0.01,
            optimizer: Optional[torch.optim.Optimizer] = None):
        """run1.py:197-241: AdamW(lr=0.01) over the zipped boundary / collocation loaders."""
        opt = optimizer or torch.optim.AdamW(self.parameters(), lr=lr)
        for _ in range(epochs):
            for bc_batch, pde_batch in zip(dataloader1, dataloader2):
                _, pde_loss, bc_loss = self.train_step(opt, u1, u2, u_s1, u_s2, bc_batch, pde_batch)
            self.pde_losses.append(pde_loss)
            self.bc_losses.append(bc_loss)
        return self
