"""Scratch: 12-jet weight-gradient kernel alone on controlled inputs (not collected)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pde_amd import functional as K
from pde_amd.kan import KANLinear

torch.manual_seed(0)
I, Oo, G = 64, 64, 10
N = int(os.environ.get("N", 16))
lay = KANLinear(I, Oo, grid_size=G).cuda()
params = [lay.base_weight, lay.spline_weight, lay.spline_scaler]


def grads(x, gy, debug):
    os.environ["PDEKAN_J12_LAYER_DEBUG"] = "1" if debug else "0"
    for p in params:
        p.grad = None
    xj = x.clone().requires_grad_(False)
    y = K.kan_layer_jet(xj, lay.grid, lay.base_weight, lay.spline_weight, lay.spline_scaler, 1, G, 3, 6, 5)
    y.backward(gy)
    return [p.grad.clone() for p in params]


def report(tag, x, gy):
    ref = grads(x, gy, False)
    got = grads(x, gy, True)
    d = (ref[1] - got[1]).abs()
    bad = [i for i, v in enumerate(d.amax(dim=(0, 2)).tolist()) if v > 1e-6 * ref[1].abs().max().item()]
    print(f"{tag}: base err {(ref[0] - got[0]).abs().max().item():.2e}  spline err {d.max().item():.2e} of {ref[1].abs().max().item():.2e}  bad features {bad}")
    return ref, got


mode = os.environ.get("MODE", "all")
x = torch.randn(N, I, 12, device="cuda") * 0.5
x[..., 0] = torch.rand(N, I, device="cuda") * 1.8 - 0.9
gy = torch.randn(N, Oo, 12, device="cuda")
report("random", x, gy)
for slot in range(12):
    g1 = torch.zeros_like(gy); g1[..., slot] = gy[..., slot]
    report(f"gy slot {slot} only", x, g1)
for p in range(0, N, max(1, N // 16)):
    x1 = x.clone(); g1 = torch.zeros_like(gy); g1[p] = gy[p]
    report(f"gy point {p} only", x1, g1)
x2 = x.clone(); x2[..., 0] = 0.3
report("all values 0.3", x2, gy)
x3 = x.clone(); x3[:, :, 0] = x[:1, :, 0]
report("same value row for every point", x3, gy)
