"""Accuracy references for a trained Black-Scholes PINN: the closed-form European
call (B-S.py:5-17) and the implicit finite-difference grid of FDM.py:18-63.

TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product or the
examples).  Both are restated in numpy; the finite-difference solver follows
the reference scheme exactly - fully implicit in time, central in S, S_max =
3K, the boundary columns held at 0 and S_max - K and NOT folded into the
right-hand side (FDM.py:24-28, :55-62) - but solves the tridiagonal system
with a banded solver instead of forming the dense inverse (FDM.py:57).
Parity pinned by the two references agreeing with each other
(tests/test_bs_accuracy.py); the reference ships no fixture for them.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.linalg import solve_banded


def closed_form_call(S, K, T, r, sigma):
    """B-S.py:5-17 (call branch).  T = time to expiry; S, T broadcast."""
    S = np.asarray(S, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    Ts = np.maximum(T, 1e-300)
    Ss = np.maximum(S, 1e-300)
    d1 = (np.log(Ss / K) + (r + 0.5 * sigma ** 2) * Ts) / (sigma * np.sqrt(Ts))
    d2 = d1 - sigma * np.sqrt(Ts)
    cdf = np.vectorize(lambda z: 0.5 * (1.0 + math.erf(z / math.sqrt(2.0))))
    price = Ss * cdf(d1) - K * np.exp(-r * Ts) * cdf(d2)
    return np.where(T <= 0, np.maximum(S - K, 0.0), price)


def fdm_call_grid(T: float, r: float, v: float, M: int, K: float, fold_boundary: bool = False) -> np.ndarray:
    """FDM.py:18-63.  Returns f[M+1, M+1]: row i = time level i*T/M (row M =
    expiry), column j = asset price j * 3K / M.

    The reference leaves the known boundary columns out of the right-hand side
    (FDM.py:58-62), which pulls the solution towards 0 next to S_max and, step by
    step, further in; ``fold_boundary=True`` adds the missing term (the textbook
    scheme) and is what the PINN is compared with away from the money."""
    dT = T / M
    S_max = 3.0 * K
    dS = S_max / M
    f = np.zeros((M + 1, M + 1))
    f[M, :] = np.maximum(dS * np.arange(M + 1) - K, 0.0)          # FDM.py:26-27
    f[:, M] = S_max - K                                            # FDM.py:28
    f[:, 0] = 0.0
    j = np.arange(1, M, dtype=np.float64)
    vj2 = (v * j) ** 2                                             # FDM.py:31-36
    a = 0.5 * dT * (r * j - vj2)
    b = 1.0 + dT * (vj2 + r)
    c = -0.5 * dT * (r * j + vj2)
    ab = np.zeros((3, M - 1))
    ab[0, 1:] = c[:-1]
    ab[1, :] = b
    ab[2, :-1] = a[1:]
    for i in range(M, 0, -1):                                      # FDM.py:58-62
        rhs = f[i, 1:M].copy()
        if fold_boundary:
            rhs[-1] -= c[-1] * f[i - 1, M]
            rhs[0] -= a[0] * f[i - 1, 0]
        f[i - 1, 1:M] = solve_banded((1, 1), ab, rhs)
    return f
