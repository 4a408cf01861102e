"""PyTorch-side helpers for the layer's NON-hot methods (initialisation, grid
adaptation, coefficient fitting).  They run rarely and on small tensors, so
they stay ordinary tensor code; the forward/backward hot path never calls them.
"""
from __future__ import annotations

import torch


def extended_knots(in_features: int, grid_size: int, spline_order: int, grid_range) -> torch.Tensor:
    """``[in_features, grid_size + 2*spline_order + 1]`` uniform knots with
    ``spline_order`` extra intervals on each side (efficient_kan/kan.py:27-35).
    Built as integer steps times h plus the lower end, all in the default
    dtype, so seeded models carry bit-identical ``grid`` buffers."""
    lo, hi = grid_range
    h = (hi - lo) / grid_size
    steps = torch.arange(-spline_order, grid_size + spline_order + 1)
    return (steps * h + lo).expand(in_features, -1).contiguous()


def dense_bsplines(x: torch.Tensor, grid: torch.Tensor, spline_order: int) -> torch.Tensor:
    """Every order-``spline_order`` basis of every feature,
    ``[batch, in_features, grid_size + spline_order]`` (kan.py:78-111).

    Level 0 is the indicator of each half-open knot interval; level p blends
    neighbouring level p-1 bases with weights that are ratios of knot
    distances.  Used only by ``curve2coeff`` / ``update_grid`` / ``b_splines``.
    """
    if x.dim() != 2 or x.size(1) != grid.size(0):
        raise AssertionError("expected x of shape (batch, in_features)")
    xs = x.unsqueeze(-1)
    level = ((xs >= grid[:, :-1]) & (xs < grid[:, 1:])).to(x.dtype)
    for p in range(1, spline_order + 1):
        span_lo = grid[:, p:-1] - grid[:, : -(p + 1)]
        span_hi = grid[:, p + 1:] - grid[:, 1:(-p)]
        level = (xs - grid[:, : -(p + 1)]) / span_lo * level[:, :, :-1] + \
                (grid[:, p + 1:] - xs) / span_hi * level[:, :, 1:]
    return level.contiguous()


def fit_coefficients(x: torch.Tensor, y: torch.Tensor, grid: torch.Tensor, spline_order: int) -> torch.Tensor:
    """Least-squares spline coefficients through samples (kan.py:113-143).

    x ``[batch, in]``, y ``[batch, in, out]`` -> ``[out, in, coeff]``: one
    small lstsq per input feature, batched."""
    design = dense_bsplines(x, grid, spline_order).transpose(0, 1)      # [in, batch, coeff]
    rhs = y.transpose(0, 1)                                             # [in, batch, out]
    sol = torch.linalg.lstsq(design, rhs).solution                      # [in, coeff, out]
    return sol.permute(2, 0, 1).contiguous()
