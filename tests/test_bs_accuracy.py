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


def test_market_quotes_fixture_is_the_references_file():
    """tests/golden/data_quotes.npz holds the 72 rows of the reference's data.csv (K, X, T, V): one strike, days to
    expiry counting down, positive prices - and the closed form with the scripts' constants (run1.py:126-127,
    429-434) prices them within the spread the scripts report against (tens of percent)."""
    import numpy as np
    from conftest import load_golden
    q = load_golden("data_quotes")
    assert all(q[k].shape == (72,) for k in ("K", "X", "T", "V"))
    assert np.all(q["K"] == 2.411) and np.all(q["V"] > 0) and q["T"].max() == 121
    model = closed_form_call(q["X"], 2.411, q["T"] / 365.0, 0.025610, 0.165856529)
    assert np.median(np.abs(model - q["V"]) / q["V"]) < 0.5
