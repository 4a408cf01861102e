"""Time one PI-DeepONet training step (run1.py:217-233 at the script's sizes: 3000 sensor rows, batch 100, networks of
run1.py:478-482) four ways: one library call or layer by layer, eagerly or replayed from a HIP graph.

    python scripts/time_deeponet_step.py [--sensors 3000] [--batch 100] [--steps 300]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(m, fused, graph):
    from pde_amd import KAN
    from pde_amd.deeponet import PI_DeepONet
    torch.manual_seed(0)
    ident = torch.nn.Identity
    net = PI_DeepONet(KAN([3, 2, 1], base_activation=ident), KAN([3, 2, 1], base_activation=ident),
                      KAN([m, 2, 1], base_activation=ident), KAN([2, 6, 1], base_activation=ident)).cuda()
    net.fused = fused
    # (fused=True: the update of all tensors in one launch; same arithmetic as the default, which takes a dozen launches)
    opt = torch.optim.AdamW(net.parameters(), lr=0.01, capturable=graph, fused=True)
    return net, opt


class _NoUpdate:
    def step(self):
        pass

    def zero_grad(self):
        pass


def run(m, nb, steps, fused, graph, update=True):
    net, opt = build(m, fused, graph)
    if not update:                 # losses and gradients only
        params = list(net.parameters())
        opt = _NoUpdate()
        opt.zero_grad = lambda: [setattr(p, "grad", None) for p in params]
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.rand(*s, generator=g).cuda()
    u1, u2 = r(m, 3), r(m, 3)
    bc, pde = [r(nb), r(nb), r(nb)], [r(nb), r(nb), torch.zeros(nb).cuda()]
    step = lambda: net.train_step(opt, u1, u2, None, None, bc, pde)
    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        step = gr.replay
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(steps):
        step()
    b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    return {"gpu_ms": a.elapsed_time(b) / steps, "wall_ms": wall}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sensors", type=int, default=3000)
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    out = {"sensors": args.sensors, "batch": args.batch, "steps": args.steps}
    for fused in (True, False):
        for graph in (True, False):
            key = ("one_call" if fused else "layerwise") + ("_graph" if graph else "_eager")
            out[key] = run(args.sensors, args.batch, args.steps, fused, graph)
    out["one_call_graph_no_update"] = run(args.sensors, args.batch, args.steps, True, True, update=False)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
