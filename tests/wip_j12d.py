"""Scratch: where the 12-jet weight gradients differ (not collected)."""
import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_parity as T
from pde_amd.pinn import BlackScholesStep

widths, N = [6, 64, 64, 1], int(os.environ.get("N", 1008))
m = T._random_model(widths, 10, 7)
x = torch.rand(N, 6) * 0.9 + 0.05
mc = copy.deepcopy(m).cuda()
out = {}
for flag in ("1", "0"):
    os.environ["PDEKAN_DISABLE_F16"] = flag
    step = BlackScholesStep(mc)
    step(x.cuda())
    out[flag] = [p.grad.clone() for p in mc.parameters()]
names = [n for n, _ in mc.named_parameters()]
for n, a, b in zip(names, out["1"], out["0"]):
    if n != "layers.1.spline_weight":
        continue
    d = (a - b).abs()
    print(n, tuple(a.shape), "max", a.abs().max().item())
    print(" err by column c:", [f"{v:.1e}" for v in d.amax(dim=(0, 1)).tolist()])
    print(" ref by column c:", [f"{v:.1e}" for v in a.abs().amax(dim=(0, 1)).tolist()])
    print(" bad features:", [i for i, v in enumerate(d.amax(dim=(0, 2)).tolist()) if v > 1e-8])
    print(" bad outputs:", [i for i, v in enumerate(d.amax(dim=(1, 2)).tolist()) if v > 1e-8][:70])
    print(" err by output o (first 8):", [f"{v:.1e}" for v in d.amax(dim=(1, 2)).tolist()[:8]])
    for i in (6, 7, 14, 15, 22, 23):
        print(" feature", i, "bad cols", [c for c, v in enumerate(d[:, i].amax(dim=0).tolist()) if v > 1e-8], "nonzero cols", [c for c, v in enumerate(a[:, i].abs().amax(dim=0).tolist()) if v > 0])
    i = d.amax(dim=(0, 2)).argmax().item()
    print(" feature", i, "ref row o=0:", a[0, i].tolist())
    print(" feature", i, "f16 row o=0:", b[0, i].tolist())
jet = mc.forward_jet(x.cuda(), 5)
