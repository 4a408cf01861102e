"""Scratch check of the 12-jet matrix kernels against an fp64 evaluation (not collected)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pde_amd import functional as K
from pde_amd.kan import KANLinear
from oracle import kan_oracle as O


def main():
    torch.manual_seed(0)
    I, Oo, G, N = int(os.environ.get("I", 64)), 64, 10, 1000
    lay = KANLinear(I, Oo, grid_size=G).cuda()
    x = torch.randn(N, I, 12, device="cuda") * 0.5
    x[..., 0].clamp_(-0.95, 0.95)
    args = (lay.grid, lay.base_weight, lay.spline_weight, lay.spline_scaler, 1, G, 3)
    os.environ["PDEKAN_DISABLE_F16"] = "1"
    y_ref = K.kan_layer_jet(x, *args, 6, 5).clone()
    os.environ["PDEKAN_DISABLE_F16"] = "0"
    y = K.kan_layer_jet(x, *args, 6, 5).clone()
    L64 = O.layers_from_state_dict({"layers.0." + k: v.cpu() for k, v in lay.state_dict().items()}, dtype=torch.float64)
    y64 = O.layer_jet(x.cpu().double(), L64[0], "silu", 6, 5) if hasattr(O, "layer_jet") else None
    for j in range(12):
        d = (y[..., j] - y_ref[..., j]).abs().max().item()
        s = y_ref[..., j].abs().max().item()
        line = f"slot {j:2d}: max|f16-f32| {d:.3e}  max|ref| {s:.3e}  rel {d / s:.2e}"
        if y64 is not None:
            e16 = (y[..., j].cpu().double() - y64[..., j]).abs().max().item()
            e32 = (y_ref[..., j].cpu().double() - y64[..., j]).abs().max().item()
            line += f"   vs fp64: f16 {e16 / s:.2e}  f32 {e32 / s:.2e}"
        print(line)


main()
