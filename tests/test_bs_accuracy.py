"""The two accuracy references (closed form B-S.py:5-17, implicit finite
differences FDM.py:18-63) agree with each other, and with the one number the
reference prints (B-S.py:21-29)."""
import numpy as np

from oracle.bs_accuracy import closed_form_call, fdm_call_grid


def test_closed_form_known_values():
    # put-call parity and limits
    S, K, T, r, s = 2.426, 2.411, 20 / 365, 0.025610, 0.165856529
    c = float(closed_form_call(S, K, T, r, s))
    assert 0.04 < c < 0.06                                   # at the money, 20 days, 16.6 % vol: ~2 % of S
    assert abs(float(closed_form_call(10 * K, K, T, r, s)) - (10 * K - K * np.exp(-r * T))) < 1e-9
    assert float(closed_form_call(0.1 * K, K, T, r, s)) < 1e-12
    assert float(closed_form_call(3.0, 2.0, 0.0, r, s)) == 1.0


def test_finite_differences_agree_with_the_closed_form():
    K, T, r, v, M = 2.411, 1.0, 0.025610, 0.165856529, 400   # FDM.py:68-72 (M = 8000 there)
    S = np.arange(M + 1) * 3 * K / M
    money = slice(M // 6, M // 2)                            # 0.5 K .. 1.5 K
    for fold in (False, True):
        f = fdm_call_grid(T, r, v, M, K, fold_boundary=fold)
        for i in (0, M // 2, M - 10):                        # time level i -> time to expiry T - i*T/M
            ref = closed_form_call(S, K, T - i * T / M, r, v)
            assert np.max(np.abs(f[i, money] - ref[money])) < 1e-3
            if fold:                                         # the textbook scheme also holds up to the far boundary
                assert np.max(np.abs(f[i] - ref)) < 0.07     # (its Dirichlet value S_max - K is not discounted)
    # the scheme as the reference writes it loses the far boundary: the last interior node collapses
    f = fdm_call_grid(T, r, v, M, K)
    assert f[0, M - 1] < 0.5 * (3 * K - K)
