"""Data parallelism over collocation points: one process per GPU, parameters
replicated, each rank takes a contiguous shard of the points.

The loss of run1.py:148 is a mean over independent points, so a shard computes
``sum(res^2) / n_global`` and its parameter-gradient contribution; ONE
all-reduce(sum) of the flat gradient buffer (plus the scalar loss, appended to
the same message) makes every replica hold the full-batch gradient, after which
identical optimiser updates keep the replicas identical - no broadcast.  With
the ``nccl`` backend on ROCm this is RCCL over xGMI; with ``gloo`` the same
code runs on CPU tensors (used by the tests).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard ``[lo, hi)`` of ``n_total`` points: the first
    ``n_total % world`` ranks get one extra point."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_points(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(x.size(0), rank, world)
    return x[lo:hi]


def sync_replicas(tensors, dist_module, check_only: bool = False) -> None:
    """Make the replicated tensors of every rank equal to rank 0's (one broadcast of one flat message), or, with
    ``check_only``, compare them with rank 0's and raise ``RuntimeError`` on the ranks that differ.

    The message is the tensors' BYTES (uint8 views, whatever their dtypes): float64 parameters and integer buffers
    travel unrounded, and the comparison is of bit patterns, so a NaN that every rank holds is not a mismatch."""
    tensors = [t.detach() for t in tensors if t.numel() > 0]
    if not tensors:
        return
    flat = torch.cat([t.contiguous().reshape(-1).view(torch.uint8) for t in tensors])
    ref = flat.clone()
    dist_module.broadcast(ref, src=0)
    if check_only:
        bad = torch.zeros(1, dtype=torch.float32, device=flat.device)
        if not torch.equal(ref, flat):
            bad += 1
        dist_module.all_reduce(bad, op=dist_module.ReduceOp.SUM)      # every rank learns of a mismatch anywhere
        if bad.item() > 0:
            raise RuntimeError(f"pde_amd.dist: replicated tensors differ from rank 0's on {int(bad.item())} rank(s); "
                               "seed or load every rank identically, or pass sync='broadcast'")
        return
    if dist_module.get_rank() == 0:                # rank 0 holds the source: nothing to copy back
        return
    off = 0
    for t in tensors:
        nbytes = t.numel() * t.element_size()
        t.copy_(ref[off: off + nbytes].clone().view(t.dtype).view_as(t))    # (clone: a view needs an aligned offset)
        off += nbytes


class FlatAllReduce:
    """Sum ``flat_grad`` and ``loss`` over the ranks with a single collective."""

    def __init__(self, flat_grad: torch.Tensor, dist_module, grad_and_loss: Optional[torch.Tensor] = None):
        self.dist = dist_module
        self.flat_grad = flat_grad
        # one message: gradients followed by the loss.  A step that already keeps both in one buffer
        # (`grad_and_loss`, gradients first) is reduced in place; otherwise they are gathered into a copy.
        self.in_place = grad_and_loss is not None
        self.buf = grad_and_loss if self.in_place else torch.zeros(flat_grad.numel() + 1, dtype=flat_grad.dtype,
                                                                    device=flat_grad.device)

    def __call__(self, loss: torch.Tensor) -> torch.Tensor:
        n = self.flat_grad.numel()
        if not self.in_place:
            self.buf[:n].copy_(self.flat_grad)
            self.buf[n:].copy_(loss.reshape(1))
        self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.SUM)
        if not self.in_place:
            self.flat_grad.copy_(self.buf[:n])
        return self.buf[n]


class DataParallelStep:
    """Wrap a fused step (``pde_amd.pinn.BlackScholesStep`` or anything with the
    same call signature and a ``flat_grad`` buffer) for N ranks."""

    def __init__(self, step, dist_module, params=None, sync: str = "broadcast"):
        """``params``: the replicated tensors (parameters and knot buffers); default: the ``state_dict`` of
        ``step.model`` when the step has one.  Replicas only stay identical if they start identical, so at
        construction the tensors are made so: ``sync="broadcast"`` copies rank 0's values to every rank,
        ``sync="check"`` raises when a rank differs from rank 0, ``sync="none"`` trusts the caller."""
        self.step = step
        self.reduce = FlatAllReduce(step.flat_grad, dist_module, getattr(step, "grad_and_loss", None))
        if params is None:
            model = getattr(step, "model", None)
            params = list(model.state_dict().values()) if model is not None else []
        if sync not in ("broadcast", "check", "none"):
            raise ValueError(f"sync must be 'broadcast', 'check' or 'none', got {sync!r}")
        if sync != "none" and params:
            sync_replicas(params, dist_module, check_only=(sync == "check"))

    def __call__(self, x_shard: torch.Tensor, n_global: int, target: Optional[torch.Tensor] = None) -> torch.Tensor:
        loss = self.step(x_shard, target=target, n_global=n_global)
        return self.reduce(loss)
