"""Pins of the oracle that NEITHER restatement's author wrote: the reference holds stored vectors for
scd_ls_update only, so the other three solvers and update_with_missing are anchored on

  * scipy.optimize.nnls / L-BFGS-B answers to the optimisation problems the solvers iterate on (tests/golden/pins.json, made by
    tests/golden/make_pins.py -- the reference made its own case 3 the same way, tests/testthat/test-nnlm.R:39);
  * closed forms at k = 1, where one coordinate update is a ratio of two sums (src/base_algorithms.cpp:71-151), with and without
    missing values, masks and all three penalties -- evaluated here with plain numpy expressions, not with oracle code;
  * monotonicity of the penalised KL objective under Lee's multiplicative update.
CPU only."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import GOLDEN  # noqa: E402
from oracle import nnlm_oracle as npo  # noqa: E402
from oracle import ref  # noqa: E402

PINS = json.load(open(os.path.join(GOLDEN, "pins.json")))
TINY = 1e-16


def _solve(engine, H0, Wt, A, beta, iters, tol, method, missing=False):
    if engine == "c":
        return ref.update(H0, Wt, A, None, beta, iters, tol, method, missing=missing)[0]
    H = np.array(H0, dtype=float, order="F")
    (npo.update_with_missing if missing else npo.update)(H, Wt, A, None, beta, iters, tol, method)
    return H


# ---- least squares against Lawson-Hanson -------------------------------------------------------------------------
@pytest.mark.parametrize("idx", range(20))
def test_scd_ls_fixed_point_is_the_nnls_solution(idx):
    p = PINS["ls"][idx]
    W, y, x = np.array(p["W"]), np.array(p["y"]), np.array(p["x"])
    H = _solve("c", np.full((W.shape[1], 1), 0.5), W.T, y[:, None], p["beta"], 2000000, 1e-15, 1)
    assert np.max(np.abs(H[:, 0] - x)) < 1e-8 * max(1.0, np.max(x)), (idx, p["cond"])
    assert np.array_equal(H[:, 0] == 0, x == 0)  # the active set, exactly


@pytest.mark.parametrize("idx", [0, 1, 4, 5, 8, 12, 16])  # cond <= 1e2: numpy oracle in a few thousand sweeps
def test_scd_ls_numpy_restatement_reaches_the_nnls_solution(idx):
    p = PINS["ls"][idx]
    W, y, x = np.array(p["W"]), np.array(p["y"]), np.array(p["x"])
    H = _solve("numpy", np.full((W.shape[1], 1), 0.5), W.T, y[:, None], p["beta"], 20000, 1e-14, 1)
    assert np.max(np.abs(H[:, 0] - x)) < 1e-7 * max(1.0, np.max(x))


@pytest.mark.parametrize("idx", range(20))
def test_lee_ls_converges_to_the_nnls_solution(idx):
    """Multiplicative updates approach the same minimiser (slowly; zeros only in the limit)."""
    p = PINS["ls"][idx]
    W, y, x = np.array(p["W"]), np.array(p["y"]), np.array(p["x"])
    b = p["beta"]
    H = _solve("c", np.full((W.shape[1], 1), 0.5), W.T, y[:, None], b, 3000000, 1e-15, 2)[:, 0]

    def obj(h):
        return 0.5 * np.sum((W @ h - y) ** 2) + 0.5 * (b[0] - b[1]) * h @ h + 0.5 * b[1] * h.sum() ** 2 + b[2] * h.sum()
    assert obj(H) - obj(x) < 1e-9 * max(1.0, obj(x)), (idx, obj(H), obj(x))
    assert obj(H) >= obj(x) - 1e-12 * max(1.0, obj(x))  # scipy's answer IS the minimum
    if p["cond"] <= 1e2:
        assert np.max(np.abs(H - x)) < 1e-4 * max(1.0, np.max(x))


@pytest.mark.parametrize("idx", range(6))
@pytest.mark.parametrize("engine", ["c", "numpy"])
def test_update_with_missing_solves_each_columns_own_nnls(idx, engine):
    p = PINS["ls_na"][idx]
    W, A, X = np.array(p["W"]), np.array(p["A"], dtype=float), np.array(p["X"])
    H = _solve(engine, np.full(X.shape, 0.5), W.T, A, p["beta"], 20000 if engine == "numpy" else 500000, 1e-14, 1, missing=True)
    assert np.max(np.abs(H - X)) < 1e-7 * max(1.0, np.max(X))
    H2 = _solve("c", np.full(X.shape, 0.5), W.T, A, p["beta"], 2000000, 1e-15, 2, missing=True)
    assert np.max(np.abs(H2 - X)) < 2e-4 * max(1.0, np.max(X))


# ---- KL against L-BFGS-B ----------------------------------------------------------------------------------------------
def _kl_obj(W, a, b, h):
    wh = W @ h + TINY
    s = h.sum()
    return np.sum(wh - a * np.log(wh)) + 0.5 * (b[0] - b[1]) * h @ h + 0.5 * b[1] * s * s + b[2] * s


@pytest.mark.parametrize("idx", range(10))
@pytest.mark.parametrize("method", [3, 4])
def test_kl_solvers_reach_the_lbfgsb_minimum(idx, method):
    """lee_kl_update's fixed points are the KKT points of the penalised objective (its denominator is sumW + the penalty's
    gradient / h, src/base_algorithms.cpp:141).  scd_kl_update uses beta(0) as curvature only -- its gradient carries
    beta(2) + beta(1) (sum(Hj) - Hj(k)) but no beta(0) Hj(k) (src/base_algorithms.cpp:98-100) -- so ITS fixed point is the
    minimiser with beta(0) = 0: a property of the reference, pinned as such."""
    p = PINS["kl"][idx]
    W, a, b = np.array(p["W"]), np.array(p["a"]), p["beta"]
    x, fun, beff = (np.array(p["x"]), p["fun"], b) if method == 4 else (np.array(p["x_scd"]), p["fun_scd"], [0.0, b[1], b[2]])
    H = _solve("c", np.full((W.shape[1], 1), 0.5), W.T, a[:, None], b, 400000, 1e-15, method)[:, 0]
    f_or, f_sp = _kl_obj(W, a, beff, H), _kl_obj(W, a, beff, x)
    assert abs(f_sp - fun) < 1e-9 * abs(fun) + 1e-12
    assert abs(f_or - f_sp) < 1e-8 * max(1.0, abs(f_sp)), (idx, method, f_or, f_sp)
    assert np.max(np.abs(H - x)) < 2e-4 * max(1.0, np.max(x))


# ---- k = 1 closed forms -----------------------------------------------------------------------------------------------
def _k1_case(seed, na):
    rng = np.random.default_rng(seed)
    n, m = 17, 6
    w = rng.random(n) + 0.1
    A = rng.random((n, m)) + 0.05
    if na:
        A[rng.random((n, m)) < 0.25] = np.nan
    h = rng.random(m) + 0.2
    return w, A, h


@pytest.mark.parametrize("na", [False, True])
@pytest.mark.parametrize("beta", [[0, 0, 0], [0.3, 0.1, 0.2]])
@pytest.mark.parametrize("engine", ["c", "numpy"])
def test_k1_closed_forms_of_all_four_solvers(na, beta, engine):
    w, A, h = _k1_case(7 + na, na)
    fin = np.isfinite(A)
    A0 = np.where(fin, A, 0.0)
    b0, b1, b2 = beta
    got = {mth: _solve(engine, h[None, :], w[None, :], A, beta, 1, 1e-9, mth, missing=na)[0] for mth in (1, 2, 3, 4)}
    for j in range(A.shape[1]):
        r = fin[:, j]
        wj, aj, hj = w[r], A0[r, j], h[j]
        g = wj @ wj + (b0 - b1) + b1 + TINY  # the k = 1 Gram with its edits (src/update_with_missing.cpp:20-24)
        # 1: h - (g h - w.a + b2) / g, clipped
        assert np.isclose(got[1][j], max(hj - (g * hj - wj @ aj + b2) / g, 0.0), rtol=1e-13, atol=1e-15)
        # 2: h * w.a / (g h + b2)
        assert np.isclose(got[2][j], hj * (wj @ aj) / (g * hj + b2 + TINY), rtol=1e-13)
        # 3: one Newton step of the quadratic approximation (src/base_algorithms.cpp:93-101); sumH - h = 0 at k = 1
        mu = wj / (wj * hj + TINY)
        a2 = aj @ mu ** 2 + b0
        b_ = aj @ mu - wj.sum() + a2 * hj - b2
        assert np.isclose(got[3][j], max(b_ / (a2 + TINY), 0.0), rtol=1e-12, atol=1e-15)
        # 4: h * sum(w a / (w h)) / (sum w + b0 h + b2)
        assert np.isclose(got[4][j], hj * (wj @ (aj / (wj * hj + TINY))) / (wj.sum() + b0 * hj + b2), rtol=1e-13)


@pytest.mark.parametrize("engine", ["c", "numpy"])
def test_k1_masked_columns_do_not_move(engine):
    w, A, h = _k1_case(3, False)
    mask = np.zeros((1, A.shape[1]), dtype=int)
    mask[0, [1, 4]] = 1
    for mth in (1, 2, 3, 4):
        if engine == "c":
            H = ref.update(h[None, :], w[None, :], A, mask, [0.1, 0.05, 0.02], 3, 1e-9, mth)[0]
        else:
            H = np.array(h[None, :], order="F")
            npo.update(H, w[None, :], A, mask, [0.1, 0.05, 0.02], 3, 1e-9, mth)
        assert np.array_equal(H[0, [1, 4]], h[[1, 4]]) and np.all(H[0, [0, 2, 3, 5]] != h[[0, 2, 3, 5]])


# ---- monotonicity --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("na", [False, True])
def test_lee_kl_objective_never_increases(na):
    """Lee & Seung's theorem for the unpenalised KL objective: every multiplicative sweep is a descent step."""
    rng = np.random.default_rng(11)
    n, m, k = 25, 7, 4
    Wt = rng.random((k, n)) + 0.05
    A = rng.random((n, m)) + 0.02
    if na:
        A[rng.random((n, m)) < 0.15] = np.nan
    fin = np.isfinite(A)
    H = rng.random((k, m)) + 0.1

    def obj(Hc):
        wh = Wt.T @ Hc + TINY
        return float(np.sum(np.where(fin, wh - np.where(fin, A, 1.0) * np.log(wh), 0.0)))
    last = obj(H)
    for _ in range(40):
        H = ref.update(H, Wt, A, None, [0, 0, 0], 1, 1e-9, 4, missing=na)[0]
        cur = obj(H)
        assert cur <= last + 1e-12 * abs(last)
        last = cur
