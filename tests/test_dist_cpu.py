"""Data-parallel host logic on CPU with gloo, world_size 2: the sharded step
(oracle as the per-rank compute, it is the checker here) must reproduce the
single-process gradients and loss."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleStep:
    """Stand-in with BlackScholesStep's interface, computing with the CPU oracle."""

    def __init__(self, layers, act):
        from oracle import kan_oracle as O
        self.O, self.layers, self.act = O, layers, act
        n = sum(p[k].numel() for p in layers for k in ("base_weight", "spline_weight", "spline_scaler"))
        self.flat_grad = torch.zeros(n)

    def __call__(self, x, target=None, n_global=None):
        out = self.O.pinn_step(x, self.layers, self.act, target=target, n_global=n_global)
        flat = torch.cat([g[k].flatten() for g in out["grads"]
                          for k in ("base_weight", "spline_weight", "spline_scaler")])
        self.flat_grad.copy_(flat)
        return out["loss"]


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import kan_oracle as O
    from pde_amd.dist import DataParallelStep, shard_points
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    layers = O.random_layers([2, 8, 1], 5, seed=3)
    x = torch.rand(n_total, 2, generator=torch.Generator().manual_seed(4))
    dp = DataParallelStep(OracleStep(layers, "silu"), dist)
    loss = dp(shard_points(x, rank, world), n_global=n_total)
    q.put((rank, float(loss), dp.step.flat_grad.clone()))
    dist.barrier()
    dist.destroy_process_group()


def _layer_tensors(layers):
    return [p[k] for p in layers for k in ("grid", "base_weight", "spline_weight", "spline_scaler")]


def _worker_mismatched(rank, world, port, n_total, q):
    """Ranks that initialise with different seeds: 'check' must raise on every rank, 'broadcast' must repair."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import kan_oracle as O
    from pde_amd.dist import DataParallelStep, shard_points
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    layers = O.random_layers([2, 8, 1], 5, seed=3 + rank)
    step = OracleStep(layers, "silu")
    raised = False
    try:
        DataParallelStep(step, dist, params=_layer_tensors(layers), sync="check")
    except RuntimeError:
        raised = True
    # other dtypes travel as bytes: a float64 that float32 cannot hold, an integer above 2^24, a NaN every rank shares
    extras = [torch.tensor([1.0 + 1e-12 * (rank + 1)], dtype=torch.float64),
              torch.tensor([(1 << 24) + 1 + 2 * rank], dtype=torch.int64), torch.tensor([float("nan")])]
    dp = DataParallelStep(step, dist, params=_layer_tensors(layers) + extras, sync="broadcast")
    assert extras[0].item() == 1.0 + 1e-12 and extras[1].item() == (1 << 24) + 1
    DataParallelStep(step, dist, params=_layer_tensors(layers) + extras, sync="check")   # identical now, NaN included
    x = torch.rand(n_total, 2, generator=torch.Generator().manual_seed(4))
    loss = dp(shard_points(x, rank, world), n_global=n_total)
    q.put((rank, raised, float(loss), dp.step.flat_grad.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_seeded_differently_are_caught_or_repaired():
    from oracle import kan_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_total = 64
    procs = [ctx.Process(target=_worker_mismatched, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(raised for _, raised, _, _ in results)       # both ranks learn of the mismatch
    layers = O.random_layers([2, 8, 1], 5, seed=3)          # rank 0's values win
    x = torch.rand(n_total, 2, generator=torch.Generator().manual_seed(4))
    single = OracleStep(layers, "silu")
    loss = float(single(x))
    for _, _, l, g in results:
        assert abs(l - loss) <= 2e-6 * abs(loss)
        assert (g - single.flat_grad).abs().max() <= 2e-6 * single.flat_grad.abs().max()


@pytest.mark.parametrize("n_total", [64, 101])
def test_two_rank_step_equals_single_process(n_total):
    from oracle import kan_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    layers = O.random_layers([2, 8, 1], 5, seed=3)
    x = torch.rand(n_total, 2, generator=torch.Generator().manual_seed(4))
    single = OracleStep(layers, "silu")
    loss = float(single(x))
    for _, l, g in results:
        assert abs(l - loss) <= 2e-6 * abs(loss)
        assert (g - single.flat_grad).abs().max() <= 2e-6 * single.flat_grad.abs().max()
    assert torch.equal(results[0][2], results[1][2])      # replicas end up identical


def test_shard_bounds_cover_everything():
    from pde_amd.dist import shard_bounds
    for n in (1, 7, 64, 101, 1 << 20):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
