"""``torch.autograd.Function`` wrappers around the C ABI of libpdekan.so.

PyTorch is used here for device memory, streams and the autograd tape only;
every number comes out of the HIP kernels.  Tensors handed to the library must
live on a ROCm device - a CPU tensor raises instead of taking another path.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import PdekanLayer, PdekanLayerGrad

ACT_IDENTITY = 0
ACT_SILU = 1


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"pde_amd: {what} is on {t.device}; the KAN hot path runs only on a ROCm GPU "
            "(there is no CPU implementation in this package)")
    if t.dtype != torch.float32:
        raise TypeError(f"pde_amd: {what} must be float32, got {t.dtype}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def layer_struct(grid, base_weight, spline_weight, spline_scaler, act_kind, grid_size, spline_order) -> PdekanLayer:
    out_f, in_f = base_weight.shape
    return PdekanLayer(in_f, out_f, grid_size, spline_order, act_kind, 0,
                       grid.data_ptr(), base_weight.data_ptr(), spline_weight.data_ptr(),
                       0 if spline_scaler is None else spline_scaler.data_ptr())


def _workspace(nbytes: int, device) -> torch.Tensor:
    # torch's caching allocator returns 512-byte aligned blocks
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _prep(t: Optional[torch.Tensor], what: str) -> Optional[torch.Tensor]:
    if t is None:
        return None
    t = t.detach()
    _require_gpu(t, what)
    return t.contiguous()


def seed_jets(x: torch.Tensor, n_dir: int, n_second: int) -> torch.Tensor:
    """Input jets ``[N, d, J]`` of points ``x [N, d]`` (strided views welcome)."""
    _require_gpu(x, "x")
    assert x.dim() == 2
    lib = _lib.load()
    n, d = x.shape
    jet = torch.empty(n, d, 1 + n_dir + n_second, dtype=torch.float32, device=x.device)
    xd = x.detach()
    with torch.cuda.device(x.device):
        _lib.check(lib.pdekan_seed_jets(_ptr(xd), xd.stride(0), xd.stride(1), n, d, n_dir, n_second,
                                        _ptr(jet), _stream(x.device)))
    return jet


def _dense_layer(x, grid, base_weight, spline_weight, spline_scaler, act_kind, grid_size, spline_order):
    """``KANLinear.forward`` (kan.py:153-166) of ``x [N, in]`` as its defining expression around the dense-basis
    kernel: every op in it is differentiable to any order (``_BasisDense`` carries its own derivative rule)."""
    bases = _BasisDense.apply(x, grid, 0, grid_size, spline_order)
    act = torch.nn.functional.silu(x) if act_kind == ACT_SILU else x
    w = spline_weight if spline_scaler is None else spline_weight * spline_scaler.unsqueeze(-1)
    return torch.nn.functional.linear(act, base_weight) + \
        torch.nn.functional.linear(bases.reshape(x.size(0), -1), w.reshape(w.size(0), -1))


class _KanLayerJet(torch.autograd.Function):
    """y_jet = layer(x_jet); backward gives gx_jet and the parameter gradients.

    First-order backward (``loss.backward()``, run1.py:225) is the library's backward kernels.  A backward that is
    itself being recorded (``torch.autograd.grad(..., create_graph=True)``, the nested-gradient way of forming
    the s_x / s_xx of run1.py:122-123) cannot come from an opaque kernel call: for plain values (J = 1) it is then
    formed by differentiating the layer's defining expression around the dense-basis kernel, which is closed under
    differentiation of any order; for jets (J > 1) it raises - ``forward_jet`` already carries those derivatives."""

    @staticmethod
    def forward(ctx, x_jet, grid, base_weight, spline_weight, spline_scaler, act_kind, grid_size,
                spline_order, n_dir, n_second):
        lib = _lib.load()
        xj = _prep(x_jet, "input")
        g, bw, sw = _prep(grid, "grid"), _prep(base_weight, "base_weight"), _prep(spline_weight, "spline_weight")
        sc = _prep(spline_scaler, "spline_scaler")
        n, in_f, j = xj.shape
        assert j == 1 + n_dir + n_second and in_f == bw.shape[1]
        layer = layer_struct(g, bw, sw, sc, act_kind, grid_size, spline_order)
        y = torch.empty(n, bw.shape[0], j, dtype=torch.float32, device=xj.device)
        with torch.cuda.device(xj.device):
            ws = _workspace(lib.pdekan_layer_workspace_bytes(ctypes.byref(layer), n, n_dir, n_second), xj.device)
            _lib.check(lib.pdekan_layer_forward(ctypes.byref(layer), _ptr(xj), _ptr(y), n, n_dir, n_second,
                                                _ptr(ws), ws.numel(), _stream(xj.device)))
        # the inputs themselves (not detached copies): a recorded backward differentiates through them
        ctx.save_for_backward(x_jet, grid, base_weight, spline_weight,
                              spline_scaler if spline_scaler is not None else torch.empty(0))
        ctx.meta = (act_kind, grid_size, spline_order, n_dir, n_second, spline_scaler is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x_jet, grid, base_weight, spline_weight, spline_scaler = ctx.saved_tensors
        act_kind, grid_size, spline_order, n_dir, n_second, has_sc = ctx.meta
        spline_scaler = spline_scaler if has_sc else None
        if torch.is_grad_enabled():                    # this backward is being recorded (create_graph=True)
            return _KanLayerJet._recorded_backward(ctx, gy, x_jet, grid, base_weight, spline_weight, spline_scaler)
        lib = _lib.load()
        xj = _prep(x_jet, "input")
        g, bw, sw = _prep(grid, "grid"), _prep(base_weight, "base_weight"), _prep(spline_weight, "spline_weight")
        sc = _prep(spline_scaler, "spline_scaler")
        gy = _prep(gy, "grad_output")
        n = xj.shape[0]
        layer = layer_struct(g, bw, sw, sc, act_kind, grid_size, spline_order)
        need_x = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[2:5])
        gx = torch.empty_like(xj) if need_x else None
        gbw = torch.empty_like(bw) if need_w else None
        gsw = torch.empty_like(sw) if need_w else None
        gsc = torch.empty_like(sc) if (need_w and has_sc) else None
        grads = PdekanLayerGrad(0 if gbw is None else gbw.data_ptr(), 0 if gsw is None else gsw.data_ptr(),
                                0 if gsc is None else gsc.data_ptr())
        with torch.cuda.device(xj.device):
            ws = _workspace(lib.pdekan_layer_workspace_bytes(ctypes.byref(layer), n, n_dir, n_second), xj.device)
            _lib.check(lib.pdekan_layer_backward(ctypes.byref(layer), _ptr(xj), _ptr(gy), n, n_dir, n_second,
                                                 _ptr(gx), ctypes.byref(grads) if need_w else None, 0,
                                                 _ptr(ws), ws.numel(), _stream(xj.device)))
        return gx, None, gbw, gsw, gsc, None, None, None, None, None

    @staticmethod
    def _recorded_backward(ctx, gy, x_jet, grid, base_weight, spline_weight, spline_scaler):
        act_kind, grid_size, spline_order, n_dir, n_second, _ = ctx.meta
        if n_dir or n_second:
            raise RuntimeError(
                "pde_amd: a differentiable backward (create_graph=True) through a layer on jets is not provided; "
                "KAN.forward_jet already returns the input derivatives that nested gradients would form, and its "
                "first-order backward gives their parameter gradients")
        with torch.enable_grad():
            y = _dense_layer(x_jet.squeeze(-1), grid, base_weight, spline_weight, spline_scaler, act_kind, grid_size,
                             spline_order).unsqueeze(-1)
            wanted = [(i, t) for i, t in ((0, x_jet), (2, base_weight), (3, spline_weight), (4, spline_scaler))
                      if t is not None and ctx.needs_input_grad[i]]
            got = torch.autograd.grad(y, [t for _, t in wanted], gy, create_graph=True, allow_unused=True)
        out = [None] * 10
        for (i, _), g in zip(wanted, got):
            out[i] = g
        return tuple(out)


def kan_layer_jet(x_jet, grid, base_weight, spline_weight, spline_scaler, act_kind, grid_size, spline_order,
                  n_dir, n_second):
    return _KanLayerJet.apply(x_jet, grid, base_weight, spline_weight, spline_scaler, act_kind, grid_size,
                              spline_order, n_dir, n_second)


def kan_layer(x, grid, base_weight, spline_weight, spline_scaler, act_kind, grid_size, spline_order):
    """``KANLinear.forward`` on a 2-D batch ``[N, in] -> [N, out]`` with
    first-order autograd (values are jets with J = 1)."""
    _require_gpu(x, "x")
    y = _KanLayerJet.apply(x.contiguous().unsqueeze(-1), grid, base_weight, spline_weight, spline_scaler,
                           act_kind, grid_size, spline_order, 0, 0)
    return y.squeeze(-1)


class _BasisDense(torch.autograd.Function):
    """Dense B-spline bases ``[N, in, coeff]`` of ``x [N, in]`` or their ``order``-th derivative with respect to
    ``x``, as ONE elementwise kernel that knows its own derivative: the backward (and the forward-mode rule) of
    order k is order k+1 times the incoming tangent / cotangent, order 4 of a cubic is zero, and a batched call
    folds the batch into the rows.  That closes the op under ``torch.func`` transforms of any depth (``jacrev``,
    ``hessian``, ``vmap``, ``grad``), which is what lets ``KANLinear.forward`` stay on the GPU kernels under them."""

    generate_vmap_rule = False

    @staticmethod
    def forward(x, grid, order, grid_size, spline_order):
        _require_gpu(x, "x")
        lib = _lib.load()
        xd = x.detach().contiguous()
        gd = grid.detach().contiguous()
        n, in_f = xd.shape
        out = torch.empty(n, in_f, grid_size + spline_order, dtype=torch.float32, device=x.device)
        if n > 0:
            with torch.cuda.device(x.device):
                _lib.check(lib.pdekan_basis_dense(_ptr(gd), _ptr(xd), n, in_f, grid_size, spline_order, order,
                                                  _ptr(out), _stream(x.device)))
        return out

    @staticmethod
    def setup_context(ctx, inputs, output):
        x, grid, order, grid_size, spline_order = inputs
        ctx.save_for_backward(x, grid)
        ctx.save_for_forward(x, grid)
        ctx.meta = (order, grid_size, spline_order)

    @staticmethod
    def backward(ctx, gout):
        x, grid = ctx.saved_tensors
        order, grid_size, spline_order = ctx.meta
        if order >= spline_order:                      # derivative of a piecewise constant
            return torch.zeros_like(x), None, None, None, None
        nxt = _BasisDense.apply(x, grid, order + 1, grid_size, spline_order)
        return (gout * nxt).sum(-1), None, None, None, None

    @staticmethod
    def jvp(ctx, x_t, grid_t, *_):
        x, grid = ctx.saved_tensors
        order, grid_size, spline_order = ctx.meta
        if x_t is None or order >= spline_order:
            return torch.zeros(x.shape + (grid_size + spline_order,), dtype=x.dtype, device=x.device)
        return _BasisDense.apply(x, grid, order + 1, grid_size, spline_order) * x_t.unsqueeze(-1)

    @staticmethod
    def vmap(info, in_dims, x, grid, order, grid_size, spline_order):
        x_bdim, grid_bdim = in_dims[0], in_dims[1]
        if grid_bdim is not None:
            raise NotImplementedError("pde_amd: a batched knot grid under vmap is not supported")
        if x_bdim is None:
            return _BasisDense.apply(x, grid, order, grid_size, spline_order), None
        xb = x.movedim(x_bdim, 0)                      # [B, N, in]: the op is row-wise, so the batch folds into N
        b, n, in_f = xb.shape
        out = _BasisDense.apply(xb.reshape(b * n, in_f), grid, order, grid_size, spline_order)
        return out.view(b, n, in_f, -1), 0


def basis_dense(x: torch.Tensor, grid: torch.Tensor, grid_size: int, spline_order: int, order: int = 0) -> torch.Tensor:
    """``b_splines`` (kan.py:78-111) of ``x [N, in]`` on the GPU kernel, ``[N, in, grid_size + spline_order]``;
    differentiable in ``x`` to any order and usable under ``torch.func`` transforms."""
    return _BasisDense.apply(x, grid, order, grid_size, spline_order)


def fit_coefficients(x: torch.Tensor, grid: torch.Tensor, grid_size: int, spline_order: int,
                     y: Optional[torch.Tensor] = None, old_grid: Optional[torch.Tensor] = None,
                     old_spline_weight: Optional[torch.Tensor] = None,
                     old_spline_scaler: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``curve2coeff`` (kan.py:113-143) on the GPU: ``[out, in, grid_size + spline_order]`` least-squares spline
    coefficients through ``x [N, in]`` and ``y [N, in, out]`` (any strides), solved per feature in fp64 inside one
    kernel.  With ``y=None`` the samples are the edge functions of the layer given by ``old_grid`` /
    ``old_spline_weight`` / ``old_spline_scaler`` - the refit of ``update_grid`` (kan.py:173-180, :215) without
    the ``[N, in, out]`` tensor."""
    _require_gpu(x, "x")
    lib = _lib.load()
    n, in_f = x.shape
    gd = grid.detach().contiguous()
    xd = x.detach()
    if y is not None:
        yd = y.detach()
        out_f = yd.shape[2]
        ys = yd.stride()
        og = ow = osc = None
    else:
        yd, ys = None, (0, 0, 0)
        og = old_grid.detach().contiguous()
        ow = old_spline_weight.detach().contiguous()
        osc = None if old_spline_scaler is None else old_spline_scaler.detach().contiguous()
        out_f = ow.shape[0]
    coeff = torch.empty(out_f, in_f, grid_size + spline_order, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.pdekan_fit_coefficients(_ptr(gd), _ptr(xd), xd.stride(0), xd.stride(1), n, in_f, out_f, grid_size,
                                               spline_order, _ptr(yd), ys[0], ys[1], ys[2], _ptr(og), _ptr(ow),
                                               _ptr(osc), _ptr(coeff), _stream(x.device)))
    return coeff


def fit_on_device(x: torch.Tensor, grid_size: int, spline_order: int) -> bool:
    """Whether ``fit_coefficients`` takes these samples (float32 on a GPU, cubic splines, at most 20 intervals)."""
    return x.is_cuda and x.dtype == torch.float32 and spline_order == 3 and 1 <= grid_size <= 20 and x.size(0) > 0
