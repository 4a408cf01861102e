#!/usr/bin/env python3
"""Benchmark: collocation points per second of one KAN-PINN train step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2]

One step = forward with analytic input derivatives + Black-Scholes residual +
mean-squared loss + backward to every parameter gradient (one call into
libpdekan.so), + gradient all-reduce over RCCL when N > 1, + an AdamW update.
Points and weights are synthetic (seeded), resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.

Workloads:
  cfg1  KAN [2,5,1]            grid 5   10^4 points
  cfg2  KAN [2,64,64,1]        grid 10  2^20 points per GPU   (default at --gpus 1)
  cfg3  = cfg2 sharded, 2^24 points over 8 GPUs -> 2^21 per GPU (default at --gpus > 1)
  cfg4  KAN [2,256,256,256,1]  grid 20  2^22 points in total
  cfg5  KAN [6,128,128,1]      grid 10  2^22 points in total, 5 assets + time
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # dense fp32 matrix peak of MI355X (= fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 matrix peak
BF16X3_PRODUCTS = 6               # bf16 products per fp32-accurate product in the split-bf16 weight-gradient kernel
PEAK_HBM_GBS = 8000.0             # HBM3E spec

WORKLOADS = {
    "cfg1": dict(widths=[2, 5, 1], grid=5, pts_per_gpu=10_000),
    "cfg2": dict(widths=[2, 64, 64, 1], grid=10, pts_per_gpu=1 << 20),
    "cfg3": dict(widths=[2, 64, 64, 1], grid=10, pts_per_gpu=1 << 21),
    "cfg4": dict(widths=[2, 256, 256, 256, 1], grid=20, pts_total=1 << 22),
    "cfg5": dict(widths=[6, 128, 128, 1], grid=10, pts_total=1 << 22),
}
CPU_SAMPLE = {"cfg1": 10_000, "cfg2": 1 << 14, "cfg3": 1 << 14, "cfg4": 1 << 10, "cfg5": 1 << 12}


def kernel_cost(name, widths, grid, n_pts, J):
    """Algorithmic work of one launch of kernel `name` on n_pts points:
    (flops, bytes).  flops: the dense contraction 2*J*I*(C+1)*O per point;
    bytes: jets read and written, fp32, each tensor once."""
    C1 = grid + 4
    if not name.startswith("L"):
        if name == "bs_residual_loss":
            return 0.0, n_pts * 4.0 * (2 * J + widths[0])
        if name == "seed_jets":
            return 0.0, n_pts * 4.0 * widths[0] * (1 + J)
        return 0.0, 0.0
    layer = int(name[1:name.index(".")])
    I, O = widths[layer], widths[layer + 1]
    kind = name.split(".")[1]
    flops = 2.0 * J * I * C1 * O * n_pts
    if kind == "fwd":
        byts = 4.0 * J * (I + O) * n_pts
    elif kind == "bwd_input":
        byts = 4.0 * J * (2 * I + O) * n_pts
    elif kind == "bwd_weight":
        byts = 4.0 * J * (I + O) * n_pts
    elif kind == "head_fused":              # last layer fwd + residual + loss + its bwd in one pass
        flops = 3.0 * flops
        byts = 4.0 * (J * 2 * I + widths[0]) * n_pts
    else:                                   # reduce_unpack: O(parameters)
        return 0.0, 0.0
    return flops, byts


HIP_KERNEL_OF = {"bwd_weight.mfma": "bwd_weight_mfma_kernel<", "bwd_weight.mfma_bf16x3": "bwd_weight_bf16x3_kernel<", "bwd_input.sparse": "bwd_input_sparse_kernel<",
                 "fwd.sparse": "fwd_sparse_kernel<64, 4,", "head_fused": "head_bs_kernel<"}


def profiled_traffic(kname, name, n_local):
    """HBM bytes per launch of kernel `kname` from the committed rocprofv3 PMC summary
    (2 x FETCH_SIZE + WRITE_SIZE, KB; FETCH_SIZE counts half the bytes of wide reads on gfx950).
    The summary was taken on cfg2 at 2^20 points per GPU: only that workload gets a number."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_summary.txt")
    hip = HIP_KERNEL_OF.get(kname.split(".", 1)[1] if "." in kname else kname)
    if name not in ("cfg2",) or n_local != (1 << 20) or hip is None or not os.path.exists(path):
        return None
    fetch = write = None
    for line in open(path):
        if line.startswith(hip.split("<")[0]) and hip in line:
            for tok in line.split():
                if tok.startswith("FETCH_SIZE="):
                    fetch = float(tok.split("=")[1])
                if tok.startswith("WRITE_SIZE="):
                    write = float(tok.split("=")[1])
    if fetch is None or write is None:
        return None
    return (2.0 * fetch + write) * 1024.0


def cpu_baseline(name, model_sd, widths, act, steps=2):
    """The reference algorithm (dense Cox-de Boor + F.linear + nested autograd)
    on the host cores, on a bounded sample of the same workload."""
    from oracle import kan_oracle as O
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    n = CPU_SAMPLE[name]
    x = torch.rand(n, widths[0], generator=torch.Generator().manual_seed(0))
    layers = O.layers_from_state_dict(model_sd)
    O.pinn_step(x[: max(256, n // 8)], layers, act)       # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        O.pinn_step(x, layers, act)
    dt = (time.perf_counter() - t0) / steps
    return {"value": n / dt, "unit": "points/s", "cores": threads, "kind": "port",
            "sample": f"{steps} steps of {n} points of the same model and point distribution, "
                      f"{dt:.2f} s/step, PyTorch CPU fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--micro-batch", type=int, default=0, help="points per micro-batch inside a step (0 = all)")
    ap.add_argument("--base-activation", default="silu", choices=["silu", "identity"],
                    help="silu = the module default (kan.py:18); identity = what the scripts pass (run1.py:478-482)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="diagnostic only: skip the AdamW update")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank, to exercise the all-reduce path")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when its communicator comes up; stdout carries the one JSON line
        # the caller parses, so the C-level stdout points at stderr until the first collective has run
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from pde_amd import KAN, _lib
    from pde_amd.dist import DataParallelStep
    from pde_amd.pinn import BlackScholesStep

    name = args.workload or ("cfg2" if world == 1 else "cfg3")
    wl = WORKLOADS[name]
    widths, grid = wl["widths"], wl["grid"]
    n_local = wl["pts_per_gpu"] if "pts_per_gpu" in wl else wl["pts_total"] // world
    n_global = n_local * world
    d = widths[0]
    J = 2 * d

    torch.manual_seed(0)                                   # reference-compatible seeded init
    act_cls = torch.nn.SiLU if args.base_activation == "silu" else torch.nn.Identity
    model = KAN(widths, grid_size=grid, base_activation=act_cls).to(dev)
    # the same global point set whatever the rank count: rank r takes the r-th contiguous shard
    gen = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(n_local, d, generator=gen).to(dev)

    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    step = BlackScholesStep(model, micro_batch=args.micro_batch)
    dp = DataParallelStep(step, dist) if dist is not None else None
    opt = torch.optim.AdamW(model.parameters(), lr=0.01, fused=True)   # run1.py:209; one launch for all tensors

    def one_step():
        if dp is not None:
            loss = dp(x, n_global=n_global)
        else:
            loss = step(x)
        if not args.no_optimizer:
            opt.step()
        return loss

    for _ in range(args.warmup):
        one_step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms_per_step = 1e3 * elapsed / args.steps
    value = n_global * args.steps / elapsed
    loss_val = float(loss.item())

    # ---- per-kernel timing of the same step (HIP events on the launch stream) ----
    lib = _lib.load()
    prof_steps = 3
    lib.pdekan_profile_begin()
    for _ in range(prof_steps):
        step(x, n_global=n_global)
    torch.cuda.synchronize()
    lib.pdekan_profile_end()
    prof = _lib.profile_results()
    n_launch_pts = n_local if args.micro_batch <= 0 else min(args.micro_batch, n_local)
    kernels = {}
    for kname, (ms, cnt) in prof.items():
        # 12-jet layers run the 4-jet matrix / sparse kernels in 5 passes over gathered views: 4 jets per launch
        Jk = 4 if (J == 12 and (".mfma" in kname or kname.endswith(".sparse"))) else J
        fl, by = kernel_cost(kname, widths, grid, n_launch_pts, Jk)
        kernels[kname] = {"avg_ms": ms / cnt, "launches_per_step": cnt / prof_steps, "ms_per_step": ms / prof_steps,
                          "flops": fl, "bytes": by}
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    kd = kernels[dom]
    if dom.endswith(".mfma") or dom.endswith(".sparse"):
        # dense contraction flops against the fp32 peak (matrix = vector peak on this chip).  The sparse-support
        # kernels run on the vector ALUs and skip the 9 of 14 columns that are zero at a point: their dense-equivalent
        # rate can exceed what the ALUs execute; frac is priced on the flops they execute (5/14 of dense).
        live = (5.0 / (grid + 4)) if dom.endswith(".sparse") else 1.0
        ach = live * kd["flops"] / (kd["avg_ms"] * 1e-3) / 1e12
        roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": None, "avg_launch_ms": kd["avg_ms"],
                "algorithmic_flops_per_launch": live * kd["flops"]}
        if live < 1.0:
            roof["note"] = "fp32 vector ALUs (packed FMA), flops of the live columns only"
            roof["dense_equivalent_tflops"] = kd["flops"] / (kd["avg_ms"] * 1e-3) / 1e12
    elif dom.endswith(".mfma_bf16x3"):
        # fp32-accurate products made of 6 bf16 products each: price the algorithmic fp32 flops against bf16 peak / 6
        peak = PEAK_BF16_MFMA_TFLOPS / BF16X3_PRODUCTS
        ach = kd["flops"] / (kd["avg_ms"] * 1e-3) / 1e12
        roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": None, "avg_launch_ms": kd["avg_ms"],
                "algorithmic_flops_per_launch": kd["flops"],
                "note": "split-bf16 (3 pieces per operand, 6 bf16 MFMA products per fp32 product): peak = 2500 / 6"}
    else:
        ach = kd["bytes"] / (kd["avg_ms"] * 1e-3) / 1e9
        roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ach / PEAK_HBM_GBS, "traffic": None, "avg_launch_ms": kd["avg_ms"],
                "algorithmic_bytes_per_launch": kd["bytes"]}
    roof["traffic"] = profiled_traffic(dom, name, n_local)
    if roof["traffic"] is not None:
        roof["traffic_source"] = "profiles/r01_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
        roof["algorithmic_bytes_per_launch"] = kd["bytes"]
    step_flops = sum(k["flops"] * k["launches_per_step"] for k in kernels.values())
    kernel_ms = sum(k["ms_per_step"] for k in kernels.values())
    def _roof_s(kname, k):
        """Lower bound of one launch: the flops the kernel's formulation needs at its unit's peak, or its bytes."""
        peak = (PEAK_BF16_MFMA_TFLOPS / BF16X3_PRODUCTS) if kname.endswith("bf16x3") else PEAK_FP32_MFMA_TFLOPS
        need = k["flops"] * ((5.0 / (grid + 4)) if kname.endswith(".sparse") else 1.0)
        return max(need / (peak * 1e12), k["bytes"] / (PEAK_HBM_GBS * 1e9))
    step_roof_ms = 1e3 * sum(k["launches_per_step"] * _roof_s(n_, k) for n_, k in kernels.items())

    out = {
        "metric": "collocation pts/s per KAN-PINN train step",
        "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name}: KAN {widths} grid={grid} k=3 {'SiLU' if args.base_activation == 'silu' else 'Identity'}, Black-Scholes PINN step "
                               f"(fwd + residual + loss + bwd + AdamW), {n_local} points/GPU, {n_global} total",
                   "points_per_gpu": n_local, "points_total": n_global, "micro_batch": args.micro_batch,
                   "optimizer_in_step": not args.no_optimizer, "parallelism": f"dp{world}"},
        "loss": loss_val,
        "roofline": roof,
        "step_dense_tflops": step_flops / (ms_per_step * 1e-3) / 1e12,
        # whole-step roofline: sum over the kernels of max(flops of the formulation / its peak, algorithmic bytes / HBM peak)
        "step_roofline_ms": step_roof_ms,
        "step_roofline_frac": step_roof_ms / ms_per_step,
        "kernel_ms_per_step": kernel_ms,
        "kernels": {k: {"ms_per_step": round(v["ms_per_step"], 4), "launches_per_step": v["launches_per_step"]}
                    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_step"])},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(name, sd0, widths, args.base_activation)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
