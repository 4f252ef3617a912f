"""Pins the oracle (oracle/nnlm_oracle.py and oracle/nnlm_ref.c) before it is trusted:
  * the reference's own known-answer vectors (tests/testthat/test-nnlm.R:6-15,19-26,29-43);
  * the two independent restatements against each other on every method x mask x NA x reg combo;
  * the committed golden fixtures (tests/golden/*.npz);
  * the properties the reference's nnmf tests assert (tests/testthat/test-nnmf.R:5-24,67-93).
CPU only."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import GOLDEN, kat_case1, kat_case2, kat_case3, r_all_equal, relF  # noqa: E402
from oracle import nnlm_oracle as npo  # noqa: E402
from oracle import ref  # noqa: E402

sys.path.insert(0, GOLDEN)
import make_golden  # noqa: E402

ENGINES = [("c", ref.c_nnlm), ("numpy", npo.c_nnlm)]


@pytest.mark.parametrize("name,solver", ENGINES)
def test_kat_case1_vector_rhs(name, solver):
    A, b, tol = kat_case1()
    sol = solver(A, A @ b, [0, 0, 0], None, np.full((5, 1), 0.5), 10000, 1e-12, 1, 1)["coefficient"]
    assert r_all_equal(sol.ravel(), b, tol)


@pytest.mark.parametrize("name,solver", ENGINES)
def test_kat_case2_matrix_rhs(name, solver):
    A, b2, tol = kat_case2()
    sol = solver(A, A @ b2, [0, 0, 0], None, np.full((5, 2), 0.5), 10000, 1e-12, 1, 1)["coefficient"]
    assert r_all_equal(sol, b2, tol)


@pytest.mark.parametrize("name,solver", ENGINES)
def test_kat_case3_nnls_vector(name, solver):
    A2, b3, expected, tol = kat_case3()
    sol = solver(A2, A2 @ b3, [0, 0, 0], None, np.full((5, 1), 0.5), 10000, 1e-12, 1, 1)["coefficient"].ravel()
    assert not np.all(np.abs(sol - b3) < 1e-6)
    assert r_all_equal(sol, expected, tol)
    assert np.max(np.abs(sol - expected)) < 1e-12


@pytest.mark.parametrize("method", [1, 2, 3, 4])
@pytest.mark.parametrize("miss", [False, True])
@pytest.mark.parametrize("reg", [[0, 0, 0], [0.02, 0.01, 0.03]])
def test_c_and_numpy_restatements_agree(method, miss, reg):
    rng = np.random.default_rng(10 * method + miss)
    n, m, k = 30, 20, 4
    A = rng.random((n, m))
    if miss:
        A[rng.random((n, m)) < 0.1] = np.nan
    W0 = rng.random((n, k)) * 0.01
    H0 = rng.random((k, m)) * 0.01
    Wm = rng.random((n, k)) < 0.1
    Hm = rng.random((k, m)) < 0.1
    args = (A, k, W0, H0, Wm, Hm, reg, reg[::-1], 5, -1.0, 1, 0, True, 4, 1e-9, method, 2)
    a, b = ref.c_nnmf(*args), npo.c_nnmf(*args)
    assert relF(a["W"], b["W"]) < 1e-10 and relF(a["H"], b["H"]) < 1e-10
    for key in ("mse_error", "mkl_error", "target_error"):
        assert np.allclose(a[key], b[key], rtol=1e-11, atol=1e-13)
    assert np.array_equal(a["average_epoch"], b["average_epoch"])
    assert a["n_iteration"] == b["n_iteration"] and len(a["mse_error"]) == len(b["mse_error"]) == 3
    assert np.all(a["W"][Wm] == W0[Wm]) and np.all(a["H"][Hm] == H0[Hm])  # masked entries never move


def test_golden_halfstep_fixture_matches_oracle():
    z = np.load(os.path.join(GOLDEN, "halfstep.npz"))
    keys = sorted(k[:-5] for k in z.files if k.endswith("_meta"))
    assert len(keys) == 32
    for key in keys:
        seed, n, m, k, inner = (int(v) for v in z[key + "_meta"])
        _, method, km, na, r = key.split("_")
        A, Wt, H, mask = make_golden.halfstep_inputs(seed, n, m, k, int(km[1:]), int(na[2:]))
        reg = [0.0, 0.0, 0.0] if r == "r0" else [0.02, 0.01, 0.03]
        mth = int(method[1:])
        Hn, it = ref.update(H, Wt, A, mask, reg, inner, 1e-9, mth)
        assert np.array_equal(Hn, z[key + "_H"]) and it == int(z[key + "_it"])
        H2 = np.array(H, copy=True)
        upd = npo.update_with_missing if int(na[2:]) else npo.update
        it2 = upd(H2, Wt, A, None if mask is None else mask.astype(int), reg, inner, 1e-9, mth)
        assert relF(H2, z[key + "_H"]) < 1e-11 and it2 == it


def test_golden_driver_fixture_matches_oracle():
    z = np.load(os.path.join(GOLDEN, "driver.npz"))
    A, W0, H0 = make_golden.driver_inputs(20250928, 200, 100, 5)
    a = z["cfg1_scd_mse_args"]
    r = ref.c_nnmf(A, 5, W0, H0, None, None, list(a[4:7]), list(a[7:10]), int(a[3]), -1.0, 1, 0, True, int(a[1]), 1e-9,
                   int(a[0]), int(a[2]))
    assert np.array_equal(r["W"], z["cfg1_scd_mse_W"]) and np.array_equal(r["mse_error"], z["cfg1_scd_mse_mse_error"])
    assert r["n_iteration"] == int(z["cfg1_scd_mse_n_iteration"]) == 12
    assert len(r["mse_error"]) == 7  # i = 0,2,..,10 plus the final block: (12-1) % 2 != 0


# ---- properties asserted by the reference's tests/testthat/test-nnmf.R -------------------------------
def _exact_rank(seed=234, n=50, m=10, k=3):
    rng = np.random.default_rng(seed)
    W, H = rng.random((n, k)), rng.random((k, m))
    return W, H, W @ H


@pytest.mark.parametrize("method,max_iter,rel_tol,tol", [(1, 10000, 1e-8, 1.5e-8), (3, 2000, 1e-8, 1e-6),
                                                          (2, 10000, 1e-8, 1e-6), (4, 10000, 1e-6, 1e-3)])
def test_exact_rank_recovery(method, max_iter, rel_tol, tol):
    """test-nnmf.R:14-24: W %*% H reproduces an exactly rank-3 A for all four method x loss combos."""
    _, _, A = _exact_rank()
    rng = np.random.default_rng(123)
    inner = 50 if method < 3 else 1
    r = ref.c_nnmf(A, 3, 0.01 * rng.random((50, 3)), 0.01 * rng.random((3, 10)), None, None, [0, 0, 0], [0, 0, 0], max_iter,
                   rel_tol, 1, 0, False, inner, 1e-9, method, int(100 / inner))
    assert np.all(r["W"] >= 0) and np.all(r["H"] >= 0)
    assert r_all_equal(r["W"] @ r["H"], A, tol)


def test_mask_zero_and_na_recovery():
    """test-nnmf.R:67-85: masked entries stay exactly 0; A[1,1] = NA is imputed."""
    rng = np.random.default_rng(987)
    n, m, k = 50, 10, 3
    W, H = rng.random((n, k)), rng.random((k, m))
    Wm, Hm = rng.random((n, k)) < 0.2, rng.random((k, m)) < 0.1
    W[Wm] = 0
    H[Hm] = 0
    A = W @ H
    A0 = A.copy()
    A[0, 0] = np.nan
    Wi, Hi = 0.01 * rng.random((n, k)), 0.01 * rng.random((k, m))
    Wi[Wm] = 0
    Hi[Hm] = 0
    r = ref.c_nnmf(A, k, Wi, Hi, Wm, Hm, [0, 0, 0], [0, 0, 0], 10000, 1e-8, 1, 0, False, 50, 1e-9, 1, 2)
    assert np.all(r["W"] >= 0) and np.all(r["H"] >= 0)
    assert np.all(r["W"][Wm] == 0) and np.all(r["H"][Hm] == 0)
    assert r_all_equal(r["W"] @ r["H"], A0, 1.5e-8)


def test_missing_value_imputation():
    """test-nnmf.R:88-93: 10 % NA, held-out entries are recovered."""
    rng = np.random.default_rng(5)
    n, m, k = 50, 10, 3
    A = rng.random((n, k)) @ rng.random((k, m))
    ind = rng.choice(A.size, A.size // 10, replace=False)
    A2 = A.copy()
    A2.ravel()[ind] = np.nan
    r = ref.c_nnmf(A2, k, 0.01 * rng.random((n, k)), 0.01 * rng.random((k, m)), None, None, [0, 0, 0], [0, 0, 0], 10000, 1e-8,
                   1, 0, False, 50, 1e-9, 1, 2)
    assert r_all_equal((r["W"] @ r["H"]).ravel()[ind], A.ravel()[ind], 1.5e-8)


def test_driver_trace_bookkeeping_edges():
    """err_len, final error block and the unsigned (i-1) % trace of src/nnmf.cpp:53-54,164."""
    rng = np.random.default_rng(3)
    A = rng.random((12, 9))
    W0, H0 = rng.random((12, 2)), rng.random((2, 9))
    for max_iter, trace, expect in ((5, 2, 3), (4, 2, 3), (1, 1, 1), (6, 999999, 2), (3, 0, 3)):
        r = ref.c_nnmf(A, 2, W0, H0, None, None, [0, 0, 0], [0, 0, 0], max_iter, -1.0, 1, 0, True, 3, 1e-9, 1, trace)
        assert len(r["mse_error"]) == expect, (max_iter, trace)
        assert r["n_iteration"] == max_iter and r["warning"]  # rel.tol = -1 is never reached
    r = ref.c_nnmf(A, 2, W0, H0, None, None, [0, 0, 0], [0, 0, 0], 500, 1e-4, 1, 0, True, 50, 1e-9, 1, 2)
    assert r["n_iteration"] < 500 and not r["warning"]
