#!/usr/bin/env python3
"""Benchmark: collocation points per second of one KAN-PINN train step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2] [--graph]

One step = forward with analytic input derivatives + Black-Scholes residual +
mean-squared loss + backward to every parameter gradient (one call into
libpdekan.so), + gradient all-reduce over RCCL when N > 1, + an AdamW update.
Points and weights are synthetic (seeded), resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.

Workloads (the global point set is drawn once from seed 0 whatever the rank count; rank r takes the r-th
contiguous shard, so a strong-scaling workload sees the same points at every N):
  cfg1  KAN [2,5,1]            grid 5   10^4 points per GPU            (weak)
  cfg2  KAN [2,64,64,1]        grid 10  2^20 points per GPU            (weak; default at --gpus 1)
  cfg3  = cfg2 at 2^21 points per GPU: 2^24 over 8 GPUs                (weak; default at --gpus > 1)
  cfg4  KAN [2,256,256,256,1]  grid 20  2^22 points in total           (strong)
  cfg5  KAN [6,128,128,1]      grid 10  2^22 points in total, 5 assets + time (strong)
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3          # dense fp32 matrix peak of MI355X (= fp32 vector peak)
PEAK_16BIT_MFMA_TFLOPS = 2500.0   # dense bf16 / fp16 matrix peak
PEAK_HBM_GBS = 8000.0             # HBM3E spec
# 16-bit MFMA products per fp32-accurate product in the split-operand kernels
SPLIT_PRODUCTS = {"mfma_bf16x3": 6, "mfma_f16x2": 3}
# what "dtype": "f32" stands for in the line: inputs, outputs, accumulation and every vector-ALU kernel are fp32; the
# hidden-layer contractions multiply fp32 operands as two fp16 pieces each (three fp16 matrix products per fp32
# product, fp32 accumulate): 22 significant bits relative to each tensor's per-jet maximum, not elementwise fp32
ARITHMETIC = ("fp32 storage and accumulation; hidden-layer products as fp16 two-piece splits (hi*hi + hi*lo + lo*hi "
              "on the fp16 matrix cores, 22 bits relative to the per-jet maximum of each tensor); first layer, head, "
              "residual, loss and reductions in fp32 on the vector ALUs / fp32 matrix cores")

WORKLOADS = {
    "cfg1": dict(widths=[2, 5, 1], grid=5, pts_per_gpu=10_000),
    "cfg2": dict(widths=[2, 64, 64, 1], grid=10, pts_per_gpu=1 << 20),
    "cfg3": dict(widths=[2, 64, 64, 1], grid=10, pts_per_gpu=1 << 21),
    "cfg4": dict(widths=[2, 256, 256, 256, 1], grid=20, pts_total=1 << 22),
    "cfg5": dict(widths=[6, 128, 128, 1], grid=10, pts_total=1 << 22),
}
CPU_SAMPLE = {"cfg1": 10_000, "cfg2": 1 << 14, "cfg3": 1 << 14, "cfg4": 1 << 10, "cfg5": 1 << 12}


def kernel_cost(name, widths, grid, n_pts, J):
    """Algorithmic work of one launch of kernel `name` on n_pts points.

    Returns (dense_flops, formulation_flops, peak_tflops, bytes):
      dense_flops        2*J*I*(C+1)*O per point and contraction (every column of the expanded basis)
      formulation_flops  what the kernel's formulation has to execute: the dense count for the matrix-core
                         kernels, the 4 live spline columns + the base column (5 of C+1) for the kernels that
                         gather live weights on the vector ALUs (sparse, streaming, fused head)
      peak_tflops        peak of the unit that formulation runs on (fp32 vector = fp32 matrix = 157.3; the
                         split-operand kernels: 16-bit matrix peak / products per fp32 product)
      bytes              jets read and written, fp32, each tensor once."""
    C1 = grid + 4
    if not name.startswith("L"):
        if name == "bs_residual_loss":
            return 0.0, 0.0, PEAK_FP32_TFLOPS, n_pts * 4.0 * (2 * J + widths[0])
        if name == "small_step":           # whole step of a small model, thread = point: every layer forward and back
            dense = sum(3 * 2.0 * J * widths[l] * C1 * widths[l + 1] for l in range(len(widths) - 1)) * n_pts
            return dense, dense * 5.0 / C1, PEAK_FP32_TFLOPS, n_pts * 4.0 * (widths[0] + 1)
        if name == "seed_jets":
            return 0.0, 0.0, PEAK_FP32_TFLOPS, n_pts * 4.0 * widths[0] * (1 + J)
        return 0.0, 0.0, PEAK_FP32_TFLOPS, 0.0
    layer = int(name[1:name.index(".")])
    I, O = widths[layer], widths[layer + 1]
    parts = name.split(".")
    kind, impl = parts[1], (parts[2] if len(parts) > 2 else "")
    if impl == "standby":                   # launched beside the kernel that does the layer; returns at once on the device
        return 0.0, 0.0, PEAK_FP32_TFLOPS, 0.0
    dense = 2.0 * J * I * C1 * O * n_pts
    if kind == "fwd":
        byts = 4.0 * J * (I + O) * n_pts
    elif kind == "bwd_input":
        byts = 4.0 * J * (2 * I + O) * n_pts
    elif kind == "bwd_weight":
        byts = 4.0 * J * (I + O) * n_pts
    elif kind == "head_fused":              # last layer fwd + residual + loss + its bwd in one pass
        dense = 3.0 * dense
        byts = 4.0 * (J * 2 * I + widths[0]) * n_pts
        impl = "stream"
    elif kind == "jet_absmax":              # per-jet maxima of a layer's input jets or output cotangents: one read
        return 0.0, 0.0, PEAK_FP32_TFLOPS, 4.0 * J * max(I, O) * n_pts
    elif kind in ("gather_views", "scatter_views"):   # 12-jet tensor <-> 4-jet view of the widest side
        return 0.0, 0.0, PEAK_FP32_TFLOPS, 4.0 * (12 + 4) * max(I, O) * n_pts
    else:                                   # reduce_unpack: O(parameters)
        return 0.0, 0.0, PEAK_FP32_TFLOPS, 0.0
    if impl in SPLIT_PRODUCTS:
        return dense, dense, PEAK_16BIT_MFMA_TFLOPS / SPLIT_PRODUCTS[impl], byts
    if impl in ("mfma", "mfma_skinny"):
        return dense, dense, PEAK_FP32_TFLOPS, byts
    return dense, dense * 5.0 / C1, PEAK_FP32_TFLOPS, byts


def roofline_seconds(cost):
    """Lower bound of one launch and which roof gives it."""
    _, form, peak, byts = cost
    t_f, t_b = form / (peak * 1e12), byts / (PEAK_HBM_GBS * 1e9)
    return (t_f, "mfma") if t_f >= t_b else (t_b, "hbm")


HIP_KERNEL_OF = {"bwd_weight.mfma": "bwd_weight_mfma_kernel<", "bwd_weight.mfma_bf16x3": "bwd_weight_bf16x3_kernel<",
                 "bwd_weight.mfma_f16x2": "bwd_weight_f16", "bwd_input.sparse": "bwd_input_sparse_kernel<",
                 "fwd.sparse": "fwd_sparse_kernel<64, 4,", "fwd.mfma_f16x2": ("fwd_f16_kernel<", "fwd_sp_f16_kernel<"),
                 "bwd_input.mfma_f16x2": "bwd_input_f16_kernel<", "head_fused": "head_bs_kernel<"}


# the 12-jet kernels of the same names
HIP_KERNEL_OF_12 = {"fwd.mfma_f16x2": "fwd12_f16_kernel<", "bwd_input.mfma_f16x2": "bwd_input12_f16_kernel<",
                    "bwd_weight.mfma_f16x2": "bwd_weight12_f16_kernel<"}


def sources_digest():
    """sha256 over the kernel sources: ties a committed counter summary to the code it was measured on."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "pde_amd", "csrc", "*"))):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def profiled_traffic(kname, name, n_local, J=4):
    """HBM bytes per launch of kernel `kname` from a committed rocprofv3 PMC summary under profiles/
    (2 x FETCH_SIZE + WRITE_SIZE, KB; FETCH_SIZE counts half the bytes of wide reads on gfx950).  A summary
    only counts when its header names this workload, this per-GPU point count and the digest of the kernel
    sources it was taken on; otherwise the live line must not quote it."""
    key = kname.split(".", 1)[1] if "." in kname else kname
    hip = (HIP_KERNEL_OF_12.get(key) if J == 12 else None) or HIP_KERNEL_OF.get(key)
    if hip is None:
        return None, None
    hips = hip if isinstance(hip, tuple) else (hip,)       # (a layer's kernel may be one of several templates)
    want = f"# workload={name} n_local={n_local} sources={sources_digest()}"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary*.txt")), reverse=True):
        with open(path) as f:
            lines = f.read().splitlines()
        if not lines or lines[0].strip() != want:
            continue
        # several instantiations of a kernel template may be launched side by side (the ones that do not match the
        # layer's knot rows return at once): the one that did the work is the one with the longest launches
        fetch = write = None
        best = {"FETCH_SIZE=": -1.0, "WRITE_SIZE=": -1.0}
        for line in lines:
            if any(line.startswith(h.split("<")[0]) and h in line for h in hips) and "avg_ms=" in line:
                avg = float(line.split("avg_ms=")[1].split()[0])
                for tok in line.split():
                    for key in best:
                        if tok.startswith(key) and avg > best[key]:
                            best[key] = avg
                            if key == "FETCH_SIZE=":
                                fetch = float(tok.split("=")[1])
                            else:
                                write = float(tok.split("=")[1])
        if fetch is not None and write is not None:
            return (2.0 * fetch + write) * 1024.0, os.path.relpath(path, ROOT)
    return None, None


def host_threads():
    """Host cores this process may use: its CPU affinity, capped by OMP_NUM_THREADS (a one-GPU box grants 16 of
    the machine's cores and exports that)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cap = os.environ.get("OMP_NUM_THREADS")
    if cap and cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def cpu_baseline(name, model_sd, widths, act, steps=2):
    """The reference algorithm (dense Cox-de Boor + F.linear + nested autograd)
    on the host cores, on a bounded sample of the same workload."""
    from oracle import kan_oracle as O
    threads = host_threads()
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
                      f"{dt:.2f} s/step, PyTorch CPU fp32, all {threads} host cores granted to this process"}


def cpu_baseline_cfg1(act, steps=5):
    """BASELINE.json's CPU-runnable configuration (KAN [2,5,1] grid 5, 10^4 points) on the host cores."""
    from oracle import kan_oracle as O
    from pde_amd import KAN
    torch.manual_seed(0)
    wl = WORKLOADS["cfg1"]
    act_cls = torch.nn.SiLU if act == "silu" else torch.nn.Identity
    m = KAN(wl["widths"], grid_size=wl["grid"], base_activation=act_cls)
    base = cpu_baseline("cfg1", {k: v.detach().clone() for k, v in m.state_dict().items()}, wl["widths"], act, steps)
    base["workload"] = "cfg1: KAN [2,5,1] grid=5, 10000 points"
    return base


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh children of this script, one rank per GPU (the
    environment torch.distributed.run would give them), before this process has made any GPU call.  Rank 0 prints the
    one JSON line on the inherited stdout; the other ranks' stdout goes to stderr.  Returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    worst = 0
    for p in procs:
        rc = p.wait()
        if rc != 0 and worst == 0:
            worst = rc if rc > 0 else 128 - rc
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--micro-batch", type=int, default=0, help="points per micro-batch inside a step (0 = all)")
    ap.add_argument("--workspace-gb", type=float, default=0.0,
                    help="bound the library workspace: the largest micro-batch that fits is taken (0 = no bound)")
    ap.add_argument("--base-activation", default="silu", choices=["silu", "identity"],
                    help="silu = the module default (kan.py:18); identity = what the scripts pass (run1.py:478-482)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step (library call + all-reduce + AdamW) in a HIP graph and replay it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="diagnostic only: skip the AdamW update")
    ap.add_argument("--no-scaling-reference", action="store_true",
                    help="skip the short extra run at cfg3's per-GPU point count on the default one-GPU line")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank, to exercise the all-reduce path")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks as child processes even with --gpus 1 (the path `--gpus N` takes without a launcher)")
    args = ap.parse_args()

    if (args.gpus > 1 or args.spawn) and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))    # plain `python bench.py --gpus N`: one fresh child per GPU, no GPU call here
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist or args.spawn:
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
    from pde_amd.dist import DataParallelStep, shard_bounds
    from pde_amd.pinn import BlackScholesStep

    default_line = args.workload is None
    name = args.workload or ("cfg2" if world == 1 else "cfg3")
    wl = WORKLOADS[name]
    widths, grid = wl["widths"], wl["grid"]
    scaling = "weak" if "pts_per_gpu" in wl else "strong"
    n_global = wl["pts_per_gpu"] * world if scaling == "weak" else wl["pts_total"]
    d = widths[0]
    J = 2 * d

    torch.manual_seed(0)                                   # reference-compatible seeded init, the same on every rank
    act_cls = torch.nn.SiLU if args.base_activation == "silu" else torch.nn.Identity
    model = KAN(widths, grid_size=grid, base_activation=act_cls).to(dev)
    # ONE global point set from seed 0, whatever the rank count; rank r keeps the r-th contiguous shard
    lo, hi = shard_bounds(n_global, rank, world)
    x = torch.rand(n_global, d, generator=torch.Generator().manual_seed(0))[lo:hi].contiguous().to(dev)
    n_local = hi - lo

    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    if args.workspace_gb > 0:
        step = BlackScholesStep(model, workspace_budget=int(args.workspace_gb * 2**30))
    else:
        step = BlackScholesStep(model, micro_batch=args.micro_batch)
    dp = DataParallelStep(step, dist, sync="check") if dist is not None else None
    # run1.py:209; one launch for all tensors; capturable keeps the step count on the device (graph replay)
    opt = torch.optim.AdamW(model.parameters(), lr=0.01, fused=True, capturable=args.graph)

    def eager_step():
        if dp is not None:
            loss = dp(x, n_global=n_global)
        else:
            loss = step(x, n_global=n_global)
        if not args.no_optimizer:
            opt.step()
        return loss

    loss_first = float(eager_step().item())                # also allocates the workspace and the optimiser state
    one_step = eager_step
    if args.graph:
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            eager_step()

        def one_step():
            graph.replay()
            return step.loss

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
    args.micro_batch = step.micro_batch           # (chosen by the step when a workspace bound is given)
    n_launch_pts = n_local if args.micro_batch <= 0 else min(args.micro_batch, n_local)
    kernels = {}
    for kname, (ms, cnt) in prof.items():
        # 12-jet layers may run the 4-jet matrix / sparse kernels in 5 passes over gathered views: 4 jets per launch
        # (the fp16 two-piece kernels of such a layer take all 12 jets in one launch)
        Jk = 4 if (J == 12 and ((".mfma" in kname and not kname.endswith(".mfma_f16x2")) or kname.endswith(".sparse"))) else J
        cost = kernel_cost(kname, widths, grid, n_launch_pts, Jk)
        t_roof, bound = roofline_seconds(cost)
        kernels[kname] = {"avg_ms": ms / cnt, "launches_per_step": cnt / prof_steps, "ms_per_step": ms / prof_steps,
                          "dense_flops": cost[0], "flops": cost[1], "peak": cost[2], "bytes": cost[3],
                          "roof_ms": 1e3 * t_roof, "bound": bound}
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    kd = kernels[dom]
    if kd["bound"] == "mfma":
        ach = kd["flops"] / (kd["avg_ms"] * 1e-3) / 1e12
        roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": kd["peak"], "unit": "TFLOP/s",
                "frac": ach / kd["peak"], "traffic": None, "avg_launch_ms": kd["avg_ms"],
                "algorithmic_flops_per_launch": kd["flops"], "algorithmic_bytes_per_launch": kd["bytes"]}
        impl = dom.split(".")[2] if dom.count(".") >= 2 else ""
        if impl in SPLIT_PRODUCTS:
            roof["note"] = (f"fp32-accurate products out of {SPLIT_PRODUCTS[impl]} 16-bit matrix products each: dense "
                            f"algorithmic flops against {PEAK_16BIT_MFMA_TFLOPS:.0f} / {SPLIT_PRODUCTS[impl]} TFLOP/s "
                            "(the nominal peak; on non-zero operands this chip sustains 1543-1703 TFLOP/s of dense fp16 "
                            "matrix work at 1.73-2.0 GHz, profiles/r03_mfma_power.txt)")
        elif kd["flops"] < kd["dense_flops"]:
            roof["note"] = "fp32 vector ALUs (packed FMA): flops of the 5 live columns per feature only, fp32 vector peak"
            roof["dense_equivalent_tflops"] = kd["dense_flops"] / (kd["avg_ms"] * 1e-3) / 1e12
    else:
        ach = kd["bytes"] / (kd["avg_ms"] * 1e-3) / 1e9
        roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ach / PEAK_HBM_GBS, "traffic": None, "avg_launch_ms": kd["avg_ms"],
                "algorithmic_bytes_per_launch": kd["bytes"]}
    roof["traffic"], src = profiled_traffic(dom, name, n_local, J)
    if src:
        roof["traffic_source"] = f"{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on these kernel sources)"
    step_dense = sum(k["dense_flops"] * k["launches_per_step"] for k in kernels.values())
    kernel_ms = sum(k["ms_per_step"] for k in kernels.values())
    step_roof_ms = sum(k["launches_per_step"] * k["roof_ms"] for k in kernels.values())

    out = {
        "metric": "collocation pts/s per KAN-PINN train step",
        "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "arithmetic": ARITHMETIC, "data": "synthetic",
        "config": {"workload": f"{name}: KAN {widths} grid={grid} k=3 {'SiLU' if args.base_activation == 'silu' else 'Identity'}, Black-Scholes PINN step "
                               f"(fwd + residual + loss + bwd + AdamW), {n_local} points/GPU, {n_global} total",
                   "points_per_gpu": n_local, "points_total": n_global, "micro_batch": args.micro_batch,
                   "optimizer_in_step": not args.no_optimizer, "hip_graph": bool(args.graph),
                   "parallelism": f"dp{world}", "ranks": world,
                   "points": "global set from seed 0, contiguous shard per rank"},
        "loss_first_step": loss_first, "loss": loss_val,
        "roofline": roof,
        "step_dense_tflops": step_dense / (ms_per_step * 1e-3) / 1e12,
        # whole-step roofline: sum over the kernels of max(flops of the formulation / its peak, algorithmic bytes / HBM peak)
        "step_roofline_ms": step_roof_ms,
        "step_roofline_frac": step_roof_ms / ms_per_step,
        "kernel_ms_per_step": kernel_ms,
        "workspace_bytes": int(step._ws.numel()) if step._ws is not None else None,
        "kernels": {k: {"ms_per_step": round(v["ms_per_step"], 4), "launches_per_step": v["launches_per_step"],
                        "bound": v["bound"], "roof_frac": round(v["roof_ms"] / v["avg_ms"], 4) if v["avg_ms"] > 0 else None}
                    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_step"])},
    }
    # The multi-GPU series runs cfg3 (2^21 points per GPU); the default one-GPU line is cfg2 (2^20).  So that the
    # series has a one-GPU point with the same per-GPU work, the default line also times a few steps at 2^21.
    if world == 1 and default_line and not args.no_scaling_reference and not args.graph:
        n_ref = WORKLOADS["cfg3"]["pts_per_gpu"]
        x_ref = torch.rand(n_ref, d, generator=torch.Generator().manual_seed(0)).to(dev)
        for _ in range(3):
            step(x_ref, n_global=n_ref)
            opt.step()
        torch.cuda.synchronize()
        k_ref = max(5, args.steps // 5)
        t0 = time.perf_counter()
        for _ in range(k_ref):
            step(x_ref, n_global=n_ref)
            opt.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["weak_scaling_reference"] = {"workload": "cfg3 shard: 2^21 points on one GPU", "points_per_gpu": n_ref,
                                         "steps": k_ref, "ms_per_step": 1e3 * dt / k_ref, "value": n_ref * k_ref / dt}
        del x_ref
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(name, sd0, widths, args.base_activation)
        if name != "cfg1":
            out["cpu_baseline_cfg1"] = cpu_baseline_cfg1(args.base_activation)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
