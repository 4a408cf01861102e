"""Scratch: gradients of a basket model, 12-jet matrix kernels on / off (not collected)."""
import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_parity as T
from pde_amd.pinn import BlackScholesStep

for widths, N in (([6, 64, 64, 1], 1000), ([6, 128, 128, 1], 1008), ([6, 64, 64, 1], 8192), ([6, 128, 128, 1], 8192),
                  ([6, 128, 128, 1], 131072)):
    m = T._random_model(widths, 10, 7)
    x = torch.rand(N, 6) * 0.9 + 0.05
    mc = copy.deepcopy(m).cuda()
    out = {}
    for flag in ("1", "0"):
        os.environ["PDEKAN_DISABLE_F16"] = flag
        step = BlackScholesStep(mc)
        loss = step(x.cuda()).item()
        out[flag] = (loss, step.flat_grad.clone())
    off = 0
    print(widths, N, "loss", out["1"][0], out["0"][0])
    for l, lay in enumerate(mc.layers):
        for k in ("base_weight", "spline_weight", "spline_scaler"):
            n = getattr(lay, k).numel()
            a, b = out["1"][1][off:off + n], out["0"][1][off:off + n]
            print(f"   L{l}.{k:14s} max_rel {((a - b).abs().max() / a.abs().max()).item():.3e}")
            off += n
