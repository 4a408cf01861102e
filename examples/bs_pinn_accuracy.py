"""Train a KAN PINN on the Black-Scholes equation with the fused step and score it
against the closed-form European call (B-S.py:5-17).  (tests/test_gpu_deeponet.py
also scores the returned prices against the implicit finite-difference grid.)

The network sees x = S / S_max in [0, 1] and t = time to expiry in years in
[0, 1] and predicts V / K: the residual of run1.py:128,
    V_t - 1/2 sigma^2 x^2 V_xx - r x V_x + r V,
is unchanged by the two rescalings, so the library's residual applies as is.
Loss = PDE residual (fused kernel step, all parameter gradients in one call) +
w * (payoff at t = 0, V = 0 at x = 0, V = S_max - K e^{-rt} at x = 1) through the
ordinary forward/backward kernels.

    python examples/bs_pinn_accuracy.py --steps 3000
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STRIKE, RATE, SIGMA = 2.411, 0.025610, 0.165856529          # run1.py:429-434
S_MAX = 3.0 * STRIKE                                        # FDM.py:20


def closed_form_call(S, K, T, r, sigma):
    """European call, B-S.py:5-17 (T = time to expiry)."""
    S, T = np.asarray(S, np.float64), np.asarray(T, np.float64)
    d1 = (np.log(S / K) + (r + 0.5 * sigma ** 2) * T) / (sigma * np.sqrt(T))
    d2 = d1 - sigma * np.sqrt(T)
    cdf = np.vectorize(lambda z: 0.5 * (1.0 + math.erf(z / math.sqrt(2.0))))
    return S * cdf(d1) - K * np.exp(-r * T) * cdf(d2)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--points", type=int, default=1 << 14, help="collocation points per step")
    ap.add_argument("--bc-points", type=int, default=2048)
    ap.add_argument("--bc-weight", type=float, default=20.0)
    ap.add_argument("--width", type=int, default=16)
    ap.add_argument("--grid", type=int, default=10)
    ap.add_argument("--lr", type=float, default=5e-3)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)

    from pde_amd import KAN
    from pde_amd.pinn import BlackScholesStep

    dev = torch.device("cuda")
    torch.manual_seed(args.seed)
    model = KAN([2, args.width, args.width, 1], grid_size=args.grid).to(dev)
    step = BlackScholesStep(model, sigma=SIGMA, rate=RATE)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    gen = torch.Generator(device=dev).manual_seed(args.seed)
    nb = args.bc_points

    def boundary_batch():
        u = torch.rand(3, nb, generator=gen, device=dev)
        x0 = torch.stack([u[0], torch.zeros(nb, device=dev)], 1)                       # t = 0: payoff
        v0 = torch.clamp(u[0] * S_MAX - STRIKE, min=0.0) / STRIKE
        x1 = torch.stack([torch.zeros(nb, device=dev), u[1]], 1)                       # S = 0
        v1 = torch.zeros(nb, device=dev)
        x2 = torch.stack([torch.ones(nb, device=dev), u[2]], 1)                        # S = S_max
        v2 = (S_MAX - STRIKE * torch.exp(-RATE * u[2])) / STRIKE
        return torch.cat([x0, x1, x2], 0), torch.cat([v0, v1, v2], 0)

    t0 = time.time()
    history = []
    for it in range(args.steps):
        pts = torch.rand(args.points, 2, generator=gen, device=dev)
        pde_loss = step(pts)                                       # overwrites every p.grad
        xb, vb = boundary_batch()
        bc_loss = torch.mean((model(xb)[:, 0] - vb) ** 2)
        (args.bc_weight * bc_loss).backward()                      # adds to the same gradient buffers
        opt.step()
        if it % max(1, args.steps // 10) == 0 or it == args.steps - 1:
            history.append((it, pde_loss.item(), bc_loss.item()))
            print(f"step {it:5d}  pde {history[-1][1]:.3e}  bc {history[-1][2]:.3e}", flush=True)
    torch.cuda.synchronize()
    train_s = time.time() - t0

    # score on a grid around the money
    S = np.linspace(0.5 * STRIKE, 1.5 * STRIKE, 41)
    tau = np.linspace(0.05, 1.0, 20)
    Sg, Tg = np.meshgrid(S, tau)
    xt = torch.tensor(np.stack([Sg.reshape(-1) / S_MAX, Tg.reshape(-1)], 1), dtype=torch.float32, device=dev)
    with torch.no_grad():
        pred = (model(xt)[:, 0] * STRIKE).double().cpu().numpy()
    exact = closed_form_call(Sg.reshape(-1), STRIKE, Tg.reshape(-1), RATE, SIGMA)
    rmse_cf = float(np.sqrt(np.mean((pred - exact) ** 2)))
    print(f"RMSE vs closed form: {rmse_cf:.4e}   (prices up to {exact.max():.3f}); "
          f"{train_s:.1f} s for {args.steps} steps")
    return {"history": history, "rmse_closed_form": rmse_cf, "train_s": train_s,
            "S": Sg.reshape(-1), "tau": Tg.reshape(-1), "pred": pred, "strike": STRIKE, "rate": RATE, "sigma": SIGMA}


if __name__ == "__main__":
    main()
