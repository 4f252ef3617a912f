"""Parity tests proper: the HIP path (through the C ABI of libnnlm_mi355x.so) against the fp64 oracle and
the committed golden fixtures.  Needs a real MI355X: run with `pytest -m gpu`.

Tolerances: F64 mode (A, GEMMs and sweeps all fp64) is compared at 1e-10 relative Frobenius and its integer
outputs (sweep counts, n_iteration, trace lengths) must be exact; F32 mode (A and cross-product GEMMs in fp32
MFMA, everything else fp64; the KL solvers with fp32 state vectors) at 2e-5 for single least-squares half-steps and
at north_star's 1e-4 for KL half-steps and whole runs."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import GOLDEN, kat_case1, kat_case2, kat_case3, r_all_equal, relF  # noqa: E402
import nnlm_amd  # noqa: E402
from nnlm_amd import _lib, api  # noqa: E402
from oracle import ref  # noqa: E402

sys.path.insert(0, GOLDEN)
import make_golden  # noqa: E402

pytestmark = pytest.mark.gpu

PRECS = [("f64", _lib.PREC_F64, 1e-10), ("f32", _lib.PREC_F32, 2e-5)]


@pytest.fixture(autouse=True)
def _default_precision(monkeypatch):
    monkeypatch.delenv("NNLM_PRECISION", raising=False)


def hip_update(prec, H, Wt, A, mask, reg, inner, tol, method):
    """One H half-step on the GPU with the same signature as oracle ref.update()."""
    k, m = H.shape
    with nnlm_amd.Handle(0, prec) as h:
        h.set_matrix(A)
        h.set_factors(k, Wt.T.copy(), H, None, mask)
        h.half_step(1, reg, inner, tol, method)
        _, Hn = h.get_factors()
        return Hn, h.take_sweeps()


# ---- single half-steps --------------------------------------------------------------------------
@pytest.mark.parametrize("pname,prec,tol", PRECS)
@pytest.mark.parametrize("method", [1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(200, 100, 5), (257, 129, 17), (515, 131, 50), (64, 700, 64), (33, 1, 1)])
def test_half_step_matches_oracle(pname, prec, tol, method, shape):
    n, m, k = shape
    rng = np.random.default_rng(n + m + k + method)
    A = rng.random((n, m))
    W0, H0 = rng.random((n, k)), rng.random((k, m))
    reg = [0.02, 0.01, 0.03]
    inner = 5 if method < 3 else 2
    if method >= 3 and pname == "f32":
        tol = 1e-4  # F32 mode runs the KL solvers with fp32 state and v_rcp_f32 quotients (k_kl.h, kl_fast_kernel)
    if method == 1 and pname == "f32":
        # round 6: the SCD chain runs on fp32 state (k_sweep_f.h).  Its error scales with how far the half-step moves a column from its
        # start: ~5e-5 from uniformly random factors as here (the worst case), 2e-7 from a warm start (scripts/exp/sweepf_exp.hip WARM=1),
        # 7e-6 after 20 iterations at full size (tests/test_gpu_fullsize.py) -- the mode's contract is 1e-4 (BASELINE.json north_star)
        tol = 1e-4
    with nnlm_amd.Handle(0, prec) as h:
        h.set_matrix(A)
        h.set_factors(k, W0, H0)
        h.half_step(0, reg, inner, 1e-9, method)  # W half-step = update() on A^T
        W1, _ = h.get_factors()
        s1 = h.take_sweeps()
        Wt_ref, it1 = ref.update(W0.T.copy(), H0, A.T.copy(), None, reg, inner, 1e-9, method)
        assert relF(W1, Wt_ref.T) < tol
        h.half_step(1, reg, inner, 1e-9, method)
        _, H1 = h.get_factors()
        s2 = h.take_sweeps()
        H_ref, it2 = ref.update(H0, Wt_ref, A, None, reg, inner, 1e-9, method)
        assert relF(H1, H_ref) < (2e-4 if (method == 1 and pname == "f32") else 10 * tol)
        assert np.all(W1 >= 0) and np.all(H1 >= 0)
        if pname == "f64":
            assert (s1, s2) == (it1, it2)


@pytest.mark.parametrize("pname,prec,tol", PRECS)
def test_golden_halfstep_fixtures(pname, prec, tol):
    """All 32 committed half-step fixtures: 4 methods x {mask} x {NA} x {reg}."""
    z = np.load(os.path.join(GOLDEN, "halfstep.npz"))
    for key in sorted(k[:-5] for k in z.files if k.endswith("_meta")):
        seed, n, m, k, inner = (int(v) for v in z[key + "_meta"])
        _, method, km, na, r = key.split("_")
        A, Wt, H, mask = make_golden.halfstep_inputs(seed, n, m, k, int(km[1:]), int(na[2:]))
        reg = [0.0, 0.0, 0.0] if r == "r0" else [0.02, 0.01, 0.03]
        Hn, it = hip_update(prec, H, Wt, A, mask, reg, inner, 1e-9, int(method[1:]))
        assert relF(Hn, z[key + "_H"]) < tol, key
        if mask is not None:
            assert np.array_equal(Hn[mask], H[mask]), key  # masked entries are bit-identical to their input
        if pname == "f64":
            assert it == int(z[key + "_it"]), key


@pytest.mark.parametrize("missing", [False, True])
@pytest.mark.parametrize("scale_a,scale_f", [(1e20, 1e-10), (1e-6, 1e3), (3.7e5, 1.0), (1.0, 2.3e-4)])  # Grams stay >> TINY_NUM
def test_half_step_is_insensitive_to_the_magnitudes_of_A_and_the_factors(scale_a, scale_f, missing):
    """The F32 mode's split-fp16 cross products rescale A (once) and the fixed factor (every half-step) by powers of two
    (k_xprod16.h): matrices and factors far outside the fp16 range must give the same relative accuracy as O(1) data.  With missing
    entries the per-column Grams run on the same split copy of the factor rows (k_missing.h, na_gram_f16_kernel)."""
    rng = np.random.default_rng(77)
    n, m, k = 300, 200, 20
    A = scale_a * rng.random((n, m)) ** 2
    if missing:
        A[rng.random((n, m)) < 0.15] = np.nan
    W0 = np.sqrt(scale_a) / scale_f * rng.random((n, k))
    H0 = scale_f * np.sqrt(scale_a) * rng.random((k, m))
    # (F32 mode: the fp32-chain solvers' cold-start error, see test_half_step_matches_oracle)
    for prec, tol in ((_lib.PREC_F32, 1e-4), (_lib.PREC_F64, 1e-10)):
        with nnlm_amd.Handle(0, prec) as h:
            h.set_matrix(A)
            h.set_factors(k, W0, H0)
            h.half_step(0, [0, 0, 0], 5, 1e-9, 1)
            W1, _ = h.get_factors()
            h.half_step(1, [0, 0, 0], 5, 1e-9, 1)
            _, H1 = h.get_factors()
        Wt_ref, _ = ref.update(W0.T.copy(), H0, A.T.copy(), None, [0, 0, 0], 5, 1e-9, 1)
        H_ref, _ = ref.update(H0.copy(), Wt_ref, A, None, [0, 0, 0], 5, 1e-9, 1)
        assert np.all(np.isfinite(W1)) and np.all(np.isfinite(H1))
        assert relF(W1, Wt_ref.T) < tol and relF(H1, H_ref) < tol


# ---- the alternating driver ------------------------------------------------------------------------
@pytest.mark.parametrize("pname,tol", [("f64", 1e-9), ("f32", 1e-4)])
@pytest.mark.parametrize("case", ["cfg1_scd_mse", "cfg1_lee_mse", "cfg1_scd_mkl", "cfg1_lee_mkl", "cfg1_scd_mse_reg"])
def test_golden_driver_config1(monkeypatch, pname, tol, case):
    """BASELINE.json configs[0] (200 x 100, k = 5) and its method variants against the committed traces."""
    monkeypatch.setenv("NNLM_PRECISION", pname)
    z = np.load(os.path.join(GOLDEN, "driver.npz"))
    A, W0, H0 = make_golden.driver_inputs(20250928, 200, 100, 5)
    a = z[case + "_args"]
    r = nnlm_amd.c_nnmf(A, 5, W0, H0, None, None, list(a[4:7]), list(a[7:10]), int(a[3]), -1.0, 1, 0, True, int(a[1]), 1e-9,
                        int(a[0]), int(a[2]))
    assert relF(r["W"], z[case + "_W"]) < tol and relF(r["H"], z[case + "_H"]) < tol
    assert r["n_iteration"] == int(z[case + "_n_iteration"]) and r["warning"]
    for key in ("mse_error", "mkl_error", "target_error"):
        assert r[key].shape == z[f"{case}_{key}"].shape
        assert np.allclose(r[key], z[f"{case}_{key}"], rtol=max(10 * tol, 1e-8), atol=1e-12), key
    if pname == "f64":
        assert np.array_equal(r["average_epoch"], z[case + "_average_epoch"])
    else:
        assert np.allclose(r["average_epoch"], z[case + "_average_epoch"], rtol=0.05)


@pytest.mark.parametrize("pname,tol", [("f64", 1e-9), ("f32", 1e-4)])
def test_golden_driver_missing_values_and_regularisation(monkeypatch, pname, tol):
    """BASELINE.json configs[4] at small size: 10 % NA + L1/L2 (update_with_missing path)."""
    monkeypatch.setenv("NNLM_PRECISION", pname)
    z = np.load(os.path.join(GOLDEN, "driver.npz"))
    A, W0, H0 = make_golden.driver_inputs(20250928, 200, 100, 5)
    A5 = A.copy()
    A5.ravel()[np.random.default_rng(7).choice(A5.size, A5.size // 10, replace=False)] = np.nan
    with nnlm_amd.Handle(0, _lib.PREC_F64) as h:
        h.set_matrix(A5)
        info = h.matrix_info()
    assert info["any_missing"] and info["n_non_missing"] == A5.size - A5.size // 10  # exact index handling
    r = nnlm_amd.c_nnmf(A5, 5, W0, H0, None, None, [0.01, 0, 0.01], [0.01, 0, 0.01], 8, -1.0, 1, 0, True, 50, 1e-9, 1, 2)
    assert relF(r["W"], z["cfg5_na_W"]) < tol and relF(r["H"], z["cfg5_na_H"]) < tol
    assert np.allclose(r["mse_error"], z["cfg5_na_mse_error"], rtol=max(10 * tol, 1e-8))
    assert np.allclose(r["target_error"], z["cfg5_na_target_error"], rtol=max(10 * tol, 1e-8))
    if pname == "f64":
        assert np.array_equal(r["average_epoch"], z["cfg5_na_average_epoch"])


def test_default_init_and_traces_match_oracle(monkeypatch):
    """No init given: both sides draw 0.01*U(0,1) from the same stand-in generator in the reference's order
    (W first, column-major, masked entries zeroed; src/nnmf.cpp:82-98), early stop on rel.tol."""
    monkeypatch.setenv("NNLM_PRECISION", "f64")
    rng = np.random.default_rng(11)
    A = rng.random((60, 45))
    Wm, Hm = rng.random((60, 4)) < 0.1, rng.random((4, 45)) < 0.1
    args = (A, 4, None, None, Wm, Hm, [0, 0, 0], [0, 0, 0], 500, 1e-4, 1, 0, True, 50, 1e-9, 1, 2)
    r, o = nnlm_amd.c_nnmf(*args), ref.c_nnmf(*args)
    assert r["n_iteration"] == o["n_iteration"] < 500 and r["warning"] == o["warning"] is False
    assert relF(r["W"], o["W"]) < 1e-8 and relF(r["H"], o["H"]) < 1e-8
    assert np.all(r["W"][Wm] == 0) and np.all(r["H"][Hm] == 0)
    assert np.array_equal(r["average_epoch"], o["average_epoch"])
    assert np.allclose(r["target_error"], o["target_error"], rtol=1e-9)


def test_trace_bookkeeping_edges(monkeypatch):
    monkeypatch.setenv("NNLM_PRECISION", "f64")
    rng = np.random.default_rng(3)
    A = rng.random((12, 9))
    W0, H0 = rng.random((12, 2)), rng.random((2, 9))
    for max_iter, trace in ((5, 2), (4, 2), (1, 1), (6, 999999), (3, 0)):
        args = (A, 2, W0, H0, None, None, [0, 0, 0], [0, 0, 0], max_iter, -1.0, 1, 0, True, 3, 1e-9, 1, trace)
        r, o = nnlm_amd.c_nnmf(*args), ref.c_nnmf(*args)
        assert len(r["mse_error"]) == len(o["mse_error"]) and r["n_iteration"] == o["n_iteration"]
        assert np.allclose(r["mse_error"], o["mse_error"], rtol=1e-10) and np.allclose(r["mkl_error"], o["mkl_error"], rtol=1e-10)


# ---- reference known-answer vectors through nnlm_c_nnlm ---------------------------------------------
@pytest.mark.parametrize("pname", ["f64", "f32"])
def test_nnlm_known_answer_vectors(monkeypatch, pname):
    """tests/testthat/test-nnlm.R:6-15, 19-26, 29-43 with expect_equal's tolerance (1.5e-8)."""
    monkeypatch.setenv("NNLM_PRECISION", pname)
    # F32 mode rounds x and y to fp32 before the long solve (rel.tol = 1e-12): the perturbed problem's solution is
    # only good to cond(x) * 6e-8, so it is held to north_star's 1e-4; F64 mode meets expect_equal's 1.5e-8.
    tol = 1.5e-8 if pname == "f64" else 1e-4
    A, b, _ = kat_case1()
    sol = api.nnlm(A, A @ b, rng=np.random.default_rng(1))
    assert r_all_equal(sol.coefficients, b, tol)
    A, b2, _ = kat_case2()
    assert r_all_equal(api.nnlm(A, A @ b2, rng=np.random.default_rng(1)).coefficients, b2, tol)
    A2, b3, expected, _ = kat_case3()
    sol3 = api.nnlm(A2, A2 @ b3, rng=np.random.default_rng(1)).coefficients
    assert not np.all(np.abs(sol3 - b3) < 1e-6)
    assert r_all_equal(sol3, expected, tol)


def test_nnlm_with_missing_response_and_mask(monkeypatch):
    monkeypatch.setenv("NNLM_PRECISION", "f64")
    rng = np.random.default_rng(5)
    x = rng.random((40, 6))
    y = x @ rng.random((6, 3)) + 0.01 * rng.random((40, 3))
    y[rng.random(y.shape) < 0.1] = np.nan
    mask = rng.random((6, 3)) < 0.2
    b0 = (~mask).astype(float)
    r = nnlm_amd.c_nnlm(x, y, [0.01, 0, 0.001], mask, b0, 10000, 1e-12, 1, 1)
    o = ref.c_nnlm(x, y, [0.01, 0, 0.001], mask, b0, 10000, 1e-12, 1, 1)
    assert relF(r["coefficient"], o["coefficient"]) < 1e-9 and r["n_iteration"] == o["n_iteration"]
    assert np.all(r["coefficient"][mask] == 0)


# ---- properties the reference's nnmf tests assert (tests/testthat/test-nnmf.R) ---------------------
@pytest.mark.parametrize("method,loss,max_iter,rel_tol,tol", [("scd", "mse", 10000, 1e-8, 1.5e-8), ("scd", "mkl", 2000, 1e-8, 1e-6),
                                                               ("lee", "mse", 10000, 1e-8, 1e-6), ("lee", "mkl", 10000, 1e-6, 1e-3)])
def test_exact_rank_recovery(monkeypatch, method, loss, max_iter, rel_tol, tol):
    """test-nnmf.R:5-24 (n=50, m=10, k=3) through the R-interface mirror.  The reference seeds R's RNG for the default
    init; here the same-scale init is passed explicitly (Lee's slow tail makes the stopping point init dependent) and
    the run is also compared with the oracle started from the same point."""
    monkeypatch.setenv("NNLM_PRECISION", "f64")
    rng = np.random.default_rng(234)
    A = rng.random((50, 3)) @ rng.random((3, 10))
    rng = np.random.default_rng(123)
    init = {"W": 0.01 * rng.random((50, 3)), "H": 0.01 * rng.random((3, 10))}
    r = api.nnmf(A, 3, method=method, loss=loss, init=init, max_iter=max_iter, rel_tol=rel_tol, show_warning=False)
    assert np.all(r.W >= 0) and np.all(r.H >= 0)
    assert r_all_equal(r.W @ r.H, A, tol)
    args, ctx = api.prepare_nnmf(A, 3, method=method, loss=loss, init=init, max_iter=max_iter, rel_tol=rel_tol, show_warning=False)
    o = api.finish_nnmf(ref.c_nnmf(*args), ctx)
    if min(r.target_loss[-1], o.target_loss[-1]) > 1e-14:  # below that the stopping rule compares rounding noise
        assert abs(r.n_iteration - o.n_iteration) <= max(2, o.n_iteration // 50)
    assert r_all_equal(r.W @ r.H, o.W @ o.H, tol)


def test_masks_and_missing_values_properties(monkeypatch):
    """test-nnmf.R:67-93: masked entries stay exactly 0; NA entries are imputed."""
    monkeypatch.setenv("NNLM_PRECISION", "f64")
    rng = np.random.default_rng(987)
    n, m, k = 50, 10, 3
    W, H = rng.random((n, k)), rng.random((k, m))
    Wm, Hm = rng.random((n, k)) < 0.2, rng.random((k, m)) < 0.1
    W[Wm] = 0
    H[Hm] = 0
    A = W @ H
    A1 = A.copy()
    A1[0, 0] = np.nan
    r = api.nnmf(A1, k, mask={"W": Wm, "H": Hm}, max_iter=10000, rel_tol=1e-8, rng=np.random.default_rng(123), show_warning=False)
    assert np.all(r.W >= 0) and np.all(r.H >= 0) and np.all(r.W[Wm] == 0) and np.all(r.H[Hm] == 0)
    assert r_all_equal(r.W @ r.H, A, 1.5e-8)
    ind = rng.choice(A.size, A.size // 10, replace=False)
    A2 = (rng.random((n, k)) @ rng.random((k, m)))
    A3 = A2.copy()
    A3.ravel()[ind] = np.nan
    r2 = api.nnmf(A3, k, max_iter=10000, rel_tol=1e-8, rng=np.random.default_rng(567), show_warning=False)
    assert r_all_equal((r2.W @ r2.H).ravel()[ind], A2.ravel()[ind], 1.5e-8)


def test_wrapper_warning_error_and_predict(monkeypatch):
    """test-nnmf.R:52-64."""
    rng = np.random.default_rng(0)
    A = rng.random((50, 3)) @ rng.random((3, 10))
    with pytest.warns(RuntimeWarning, match="Target tolerance not reached. Try a larger max.iter."):
        api.nnmf(A, 2, alpha=0.1, beta=0, max_iter=10, rng=rng)
    with pytest.raises(api.NnlmStop):
        api.nnmf(A, 20)
    r = api.nnmf(A, 2, alpha=0.1, beta=0.01, rng=rng, show_warning=False)
    Wn = api.predict_nnmf(r, A[:4, :], which="W")
    assert Wn["coefficients"].shape == (4, 2) and np.all(Wn["coefficients"] >= 0)


# ---- real data: the vignette's algorithm comparison on data/nsclc.rda -------------------------------
@pytest.mark.parametrize("method,loss,max_iter,inner", [("scd", "mse", 30, None), ("lee", "mse", 30, None), ("scd", "mkl", 40, None),
                                                         ("lee", "mkl", 40, None), ("lee", "mse", 60, 1)])
def test_nsclc_algorithm_comparison_matches_oracle(monkeypatch, method, loss, max_iter, inner):
    """vignettes/Fast-And-Versatile-NMF.Rmd:281-296 (k = 15, uniform init, rel.tol = -1; iteration counts cut down):
    the five runs of the vignette on the reference's own data set, read with nnlm_amd.rda, against the oracle --
    factors, both error traces and the epoch trace."""
    from nnlm_amd import rda
    A = rda.load_rda(os.path.join(GOLDEN, "nsclc.rda"))["nsclc"].matrix()
    rng = np.random.default_rng(123)
    k = 15
    init = {"W": rng.random((A.shape[0], k)), "H": rng.random((k, A.shape[1]))}
    kw = dict(method=method, loss=loss, init=init, max_iter=max_iter, rel_tol=-1, show_warning=False)
    if inner is not None:
        kw["inner_max_iter"] = inner
    args, ctx = api.prepare_nnmf(A, k, **kw)
    o = api.finish_nnmf(ref.c_nnmf(*args), ctx)
    for prec, tol in (("f64", 1e-8), ("f32", 1e-4)):
        monkeypatch.setenv("NNLM_PRECISION", prec)
        r = api.nnmf(A, k, **kw)
        assert r.n_iteration == o.n_iteration == max_iter
        assert relF(r.W @ r.H, o.W @ o.H) < tol
        assert np.allclose(r.mse, o.mse, rtol=10 * tol, atol=0) and np.allclose(r.mkl, o.mkl, rtol=10 * tol, atol=1e-12)
        assert len(r.average_epochs) == len(o.average_epochs)
        if prec == "f64":
            assert np.array_equal(r.average_epochs, o.average_epochs)  # integer sweep counts


# ---- BASELINE.json configs[1] at full size ----------------------------------------------------------
def test_config2_full_size_one_iteration_vs_oracle_and_properties():
    """20000 x 10000, k = 50, MSE+SCD: one full outer iteration against the oracle (W, H within north_star's
    1e-4 relative Frobenius), then size-independent properties over a few more iterations: non-negativity,
    monotone descent of the target (SCD never increases it), exact sweep bookkeeping."""
    n, m, k = 20000, 10000, 50
    rng = np.random.default_rng(20250928)
    A = rng.random((n, m))
    W0, H0 = 0.01 * rng.random((n, k)), 0.01 * rng.random((k, m))
    z = [0.0, 0.0, 0.0]
    with nnlm_amd.Handle(0, _lib.PREC_F32) as h:
        h.set_matrix(A)
        assert h.matrix_info()["n_non_missing"] == float(n) * m
        h.set_factors(k, W0, H0)
        h.iterate(1, z, z, 50, 1e-9, 1)
        W1, H1 = h.get_factors()
        sweeps = h.take_sweeps()
        Wt_ref, it1 = ref.update(W0.T.copy(), H0, np.ascontiguousarray(A.T), None, z, 50, 1e-9, 1, missing=False)
        H_ref, it2 = ref.update(H0, Wt_ref, A, None, z, 50, 1e-9, 1, missing=False)
        assert relF(W1, Wt_ref.T) < 1e-4 and relF(H1, H_ref) < 1e-4
        assert abs(sweeps - (it1 + it2)) <= 0.001 * (it1 + it2)
        mse_prev = h.errors()[0]
        for _ in range(3):
            h.iterate(1, z, z, 50, 1e-9, 1)
            mse = h.errors()[0]
            assert mse <= mse_prev * (1 + 1e-9)
            mse_prev = mse
        W, H = h.get_factors()
        assert np.all(W >= 0) and np.all(H >= 0) and np.isfinite(W).all() and np.isfinite(H).all()
        # the device-side error block (errors_f32_kernel) against a host evaluation over the whole matrix
        host = 0.0
        for j0 in range(0, m, 1000):
            host += float(np.sum((A[:, j0:j0 + 1000] - W @ H[:, j0:j0 + 1000]) ** 2))
        host /= float(n) * m
        assert abs(host - mse_prev) < 1e-6 * host, (host, mse_prev)


# ---- multi-GPU shard arithmetic with virtual ranks on one device ---------------------------------------
@pytest.mark.parametrize("pname,prec,tol", PRECS)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_virtual_rank_partials_sum_to_the_unsharded_buffer(monkeypatch, pname, prec, tol, world):
    """(The all-reduce form of the dense half-step, nnlm_comm_set_form(NNLM_FORM_REDUCE).)  Each virtual rank computes only its slab's [Gram | cross-product]; their sum (what ncclAllReduce forms) equals
    the single-rank buffer, and the numpy Gram/cross-product of the slab nnlm_shard_range() reports."""
    rng = np.random.default_rng(world)
    n, m, k = 700, 300, 11
    A = rng.random((n, m))
    W0, H0 = rng.random((n, k)), rng.random((k, m))
    for which in (0, 1):
        with nnlm_amd.Handle(0, prec) as h:
            h.set_matrix(A)
            h.set_factors(k, W0, H0)
            Gf, Cf = h.debug_partial(which)
        Y = W0.T if which == 1 else H0
        B = A if which == 1 else A.T
        assert relF(Gf, Y @ Y.T) < 1e-12 and relF(Cf, Y @ B) < tol
        Gs, Cs = np.zeros_like(Gf), np.zeros_like(Cf)
        for rk in range(world):
            with nnlm_amd.Handle(0, prec) as h:
                h.set_matrix(A)
                h.set_factors(k, W0, H0)
                h.comm_init(None, rk, world, form="reduce")
                assert h.comm_info() == (rk, world)
                G, Cp = h.debug_partial(which)
            b, e = _lib.shard_range(n, m, prec, which, rk, world)
            assert relF(G, Y[:, b:e] @ Y[:, b:e].T) < 1e-12 and relF(Cp, Y[:, b:e] @ B[b:e, :]) < tol
            Gs += G
            Cs += Cp
        assert relF(Gs, Gf) < 1e-13 and relF(Cs, Cf) < max(tol * 1e-2, 1e-13)


@pytest.mark.parametrize("pname,prec,tol", PRECS)
@pytest.mark.parametrize("form", ["cols", "reduce"])
def test_sharded_path_with_a_real_one_rank_rccl_communicator(monkeypatch, pname, prec, tol, form):
    """The multi-GPU code path end to end on one GPU, both forms of the dense half-step: a real RCCL communicator of size 1 makes
    the handle (reduce) fold its slabs and call ncclAllReduce, (both) sweep its column slab into the packed buffer, call
    ncclAllGather and unpack.  With one rank every
    collective is the identity, so the result must be bit-identical to the plain path in the strict mode (same reduction
    orders) and equal to rounding in the f32 mode."""
    rng = np.random.default_rng(77)
    n, m, k = 300, 200, 9
    A = rng.random((n, m))
    W0, H0 = rng.random((n, k)), rng.random((k, m))
    Hm = rng.random((k, m)) < 0.1
    reg = [0.02, 0.01, 0.03]
    out = []
    for sharded in (False, True):
        with nnlm_amd.Handle(0, prec) as h:
            if sharded:
                h.comm_init(_lib.comm_unique_id(), 0, 1, form=form)
            h.set_matrix(A)
            h.set_factors(k, W0, H0, None, Hm)
            h.iterate(2, reg, reg, 6, 1e-9, 1)
            h.iterate(1, reg, reg, 3, 1e-9, 2)
            sweeps = h.take_sweeps()
            mse, kl, pen = h.errors()
            r = h.run(reg, reg, 5, -1.0, 0, False, 4, 1e-9, 1, 2)
            W, H = h.get_factors()
            out.append((W, H, sweeps, mse, kl, pen, r))
    a, b = out
    if pname == "f32":
        # the plain F32 path takes the Gram partial sums of a solved factor from the sweep kernel's LDS image (one slab per
        # workgroup of 64 columns, k_sweep_q.h), the sharded path from gram_partial_kernel (256 columns per slab): the same
        # products in another order of addition -- 1e-15 differences in the fp64 Gram (one half-step: 2e-15 in the factor),
        # which eight half-steps of coordinate descent amplify to ~1e-7 (scripts/gpu_sg_ab.py); the mode's parity bound is 1e-4
        assert relF(a[0], b[0]) < 1e-5 and relF(a[1], b[1]) < 1e-5 and abs(a[2] - b[2]) <= 2
        assert np.isclose(a[3], b[3], rtol=1e-6) and np.isclose(a[4], b[4], rtol=1e-6) and np.allclose(a[5], b[5], rtol=1e-6)
    else:
        # strict mode: since round 3 the Gram of a factor solved by SCD comes from the sweep kernel's per-workgroup partial sums on
        # the plain path (64-column slabs, folded) and in the column form (the ranks' folded sums travel behind the packed slabs),
        # from gram_partial_kernel's 256-column slabs in the reduce form -- the same products in another order of addition
        # (1e-15 in G); sweep counts stay equal
        assert relF(a[0], b[0]) < 1e-11 and relF(a[1], b[1]) < 1e-11 and a[2] == b[2]
        assert np.isclose(a[3], b[3], rtol=1e-11) and np.isclose(a[4], b[4], rtol=1e-11) and np.allclose(a[5], b[5], rtol=1e-11)
    for key in ("mse_error", "mkl_error", "target_error", "average_epoch"):
        if pname == "f32" and key != "average_epoch":
            # the plain F32 path evaluates the error sums inside the speculative cross product (xprod16_err_kernel, A rebuilt
            # from its split-fp16 copy), the sharded path with the separate kernel on the fp32 copy: same sums, other rounding
            assert np.allclose(a[6][key], b[6][key], rtol=1e-6, atol=0), key
        elif key != "average_epoch":
            assert np.allclose(a[6][key], b[6][key], rtol=1e-11, atol=0), key
        else:
            assert np.array_equal(a[6][key], b[6][key]), key
    assert np.array_equal(b[1][Hm], H0[Hm])


@pytest.mark.parametrize("pname,prec,tol", PRECS)
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("method", [1, 2])
@pytest.mark.parametrize("k", [13, 50])  # 50: the tail-block form of the fast sweep on column slabs that do not start at 0
def test_virtual_ranks_run_whole_sharded_half_steps(monkeypatch, pname, prec, tol, world, method, k):
    """Shard arithmetic of the multi-GPU path's all-reduce form with `world` virtual ranks on one device: every rank contracts its slab
    (phase 1), the host stand-in for ncclAllReduce sums the [Gram | cross-product] buffers, every rank sweeps ITS columns
    (phase 2), the stand-in for ncclAllGather distributes the packed slabs, every rank unpacks (phase 3).  All ranks must
    end with identical factors, equal to the single-rank result up to the all-reduce's summation order."""
    rng = np.random.default_rng(world + method)
    n, m = 500, 333
    A = rng.random((n, m))
    W0, H0 = rng.random((n, k)), rng.random((k, m))
    Wm = rng.random((n, k)) < 0.05
    reg = [0.02, 0.01, 0.03]
    with nnlm_amd.Handle(0, prec) as h1:
        h1.set_matrix(A)
        h1.set_factors(k, W0, H0, Wm, None)
        h1.iterate(2, reg, reg, 5, 1e-9, method)
        W_ref, H_ref = h1.get_factors()
        sw_ref = h1.take_sweeps()
        mse_ref = h1.errors()[0]
    hs = [nnlm_amd.Handle(0, prec) for _ in range(world)]
    try:
        for rk, h in enumerate(hs):
            h.comm_init(None, rk, world, form="reduce")
            h.set_matrix(A)
            h.set_factors(k, W0, H0, Wm, None)
        for _ in range(2):
            for which in (0, 1):
                for h in hs:
                    h.debug_phase(which, 1, reg, 5, 1e-9, method)
                _lib.debug_exchange(hs, which, 1)
                for h in hs:
                    h.debug_phase(which, 2, reg, 5, 1e-9, method)
                _lib.debug_exchange(hs, which, 2)
                for h in hs:
                    h.debug_phase(which, 3, reg, 5, 1e-9, method)
        res = [h.get_factors() for h in hs]
        sweeps = sum(h.take_sweeps() for h in hs)  # each rank counted its own columns
        mse = sum(h.errors()[0] for h in hs)       # each rank reduced its share of A
    finally:
        for h in hs:
            h.close()
    for W, H in res[1:]:
        assert np.array_equal(W, res[0][0]) and np.array_equal(H, res[0][1])
    t = 1e-11 if pname == "f64" else tol
    assert relF(res[0][0], W_ref) < t and relF(res[0][1], H_ref) < t
    assert np.array_equal(res[0][0][Wm], W0[Wm])
    assert sweeps == sw_ref
    assert abs(mse - mse_ref) < 1e-9 * mse_ref if pname == "f64" else abs(mse - mse_ref) < 1e-5 * mse_ref


# ---- the one-wavefront SCD sweep of the f32 mode (k_sweep_q.h): masks, columns that finish early, ragged shapes ------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(300, 101, 9), (90, 130, 13), (257, 1000, 17), (70, 50, 21), (111, 77, 26), (120, 49, 32), (64, 97, 33),
                                   (75, 200, 38), (130, 60, 44), (60, 129, 48), (400, 530, 50), (88, 65, 53), (140, 64, 58),
                                   (90, 97, 64)])  # k = 9 .. 64: every instantiation sweep_scd_q_kernel<NT, NB> has (NB = ceil(k / 4) = 3 .. 16)
@pytest.mark.parametrize("inner,itol", [(50, 1e-3), (7, 1e-9), (200, 1e-6), (0, 1e-9), (1, -1.0)])
def test_fast_sweep_with_masks_and_early_finishers(shape, inner, itol):
    n, m, k = shape
    rng = np.random.default_rng(7 * n + m + k + inner)
    A = rng.random((n, m))
    Wt = rng.random((k, n))
    H0 = rng.random((k, m)) * (rng.random((k, m)) > 0.3)  # zeros that have to stay / leave zero
    mask = rng.random((k, m)) < 0.15
    mask[:, :: 11] = True  # whole columns masked: skipped (src/update_with_missing.cpp:33)
    mask[:, 5] = False
    H0[mask] = 0.0
    reg = [0.02, 0.01, 0.03]
    H_ref, it_ref = ref.update(H0.copy(), Wt, A, mask, reg, inner, itol, 1)
    Hn, sweeps = hip_update(nnlm_amd.PREC_F32, H0.copy(), Wt, A, mask, reg, inner, itol, 1)
    assert np.all(Hn[mask] == 0.0)
    assert np.all(Hn >= 0.0)
    assert relF(Hn, H_ref) < 1e-4
    # integer sweep counts: identical decisions except for columns sitting on the tolerance
    assert abs(sweeps - it_ref) <= max(2, it_ref // 200)
    # fully masked columns are untouched, bit for bit
    assert np.array_equal(Hn[:, ::11][:, 1:], H0[:, ::11][:, 1:])


# ---- no rank limit, no contraction-length limit (k_generic.h, kl_stream_kernel) ------------------------------------------
@pytest.mark.parametrize("pname,prec,tol", PRECS)
@pytest.mark.parametrize("method", [1, 2, 3, 4])
@pytest.mark.parametrize("k", [65, 80, 128])
def test_rank_above_64_half_steps_match_oracle(pname, prec, tol, method, k):
    """The reference has no rank limit (src/update_with_missing.cpp:17-24): K = 65, 80, 128 through both half-steps, with a
    coordinate mask crossing the 64-bit word boundary and L1/L2/angle regularisation."""
    n, m = 300, 170
    rng = np.random.default_rng(k + method)
    A = rng.random((n, 10)) @ rng.random((10, m)) + 0.1 * rng.random((n, m))
    # factors on the scale of the data (W0 H0 ~ A): with W0 H0 >> A the first KL sweep drives whole columns to exactly zero and
    # the state vector y = W^T h becomes pure cancellation noise there -- a regime in which the reference's own result depends on
    # its BLAS's summation order (measured: 3e-2 between two fp64 evaluations that differ only by fused multiply-adds)
    sc = np.sqrt(A.mean() / (0.25 * k))
    W0, H0 = sc * rng.random((n, k)), sc * rng.random((k, m))
    Hm = rng.random((k, m)) < 0.1
    Hm[:, 3] = True  # a fully masked column
    H0[Hm] = 0.0
    H0[:, 3] = sc * rng.random(k)  # (masked but not zero: an all-zero column of the fixed factor is the same degenerate regime)
    reg = [0.02, 0.01, 0.03]
    inner = 4 if method < 3 else 2
    if method >= 3 and pname == "f32":
        tol = 1e-4
    with nnlm_amd.Handle(0, prec) as h:
        h.set_matrix(A)
        h.set_factors(k, W0, H0, None, Hm)
        h.half_step(0, reg, inner, 1e-9, method)
        W1, _ = h.get_factors()
        s1 = h.take_sweeps()
        Wt_ref, it1 = ref.update(W0.T.copy(), H0, A.T.copy(), None, reg, inner, 1e-9, method)
        assert relF(W1, Wt_ref.T) < tol
        h.half_step(1, reg, inner, 1e-9, method)
        _, H1 = h.get_factors()
        s2 = h.take_sweeps()
        H_ref, it2 = ref.update(H0, Wt_ref, A, Hm, reg, inner, 1e-9, method)
        assert relF(H1, H_ref) < 10 * tol
        assert np.array_equal(H1[Hm], H0[Hm])
        mse, kl, _ = h.errors()
        assert abs(mse - np.mean((A - W1 @ H1) ** 2)) < 1e-5 * mse
        if pname == "f64":
            assert (s1, s2) == (it1, it2)


@pytest.mark.parametrize("pname,tol", [("f64", 1e-9), ("f32", 1e-4)])
@pytest.mark.parametrize("method", [1, 3])
def test_rank_above_64_driver_with_missing_values(monkeypatch, pname, tol, method):
    """nnmf(k = 70) on a matrix with missing entries (update_with_missing with per-column 70 x 70 Grams) through the one-shot
    entry, traces and factors against the oracle."""
    monkeypatch.setenv("NNLM_PRECISION", pname)
    rng = np.random.default_rng(70 + method)
    n, m, k = 220, 140, 70
    A = rng.random((n, m))
    A.ravel()[rng.choice(A.size, A.size // 10, replace=False)] = np.nan
    W0, H0 = 0.1 * rng.random((n, k)), 0.1 * rng.random((k, m))
    args = (A, k, W0, H0, None, None, [0.01, 0, 0.01], [0.01, 0, 0.01], 4, -1.0, 1, 0, False, 6 if method == 1 else 1, 1e-9, method, 2)
    r, o = nnlm_amd.c_nnmf(*args), ref.c_nnmf(*args)
    assert relF(r["W"], o["W"]) < tol and relF(r["H"], o["H"]) < tol
    assert r["n_iteration"] == o["n_iteration"] and len(r["mse_error"]) == len(o["mse_error"])
    assert np.allclose(r["mse_error"], o["mse_error"], rtol=max(10 * tol, 1e-8)) and np.allclose(r["mkl_error"], o["mkl_error"], rtol=max(10 * tol, 1e-8))
    if pname == "f64":
        assert np.array_equal(r["average_epoch"], o["average_epoch"])


def test_nnlm_with_more_than_64_predictors(monkeypatch):
    """nnlm(x, y) is update() with rank = ncol(x) (src/nnlm.cpp:44-47): 90 predictors, long solve, default (fp64) precision."""
    monkeypatch.delenv("NNLM_PRECISION", raising=False)
    rng = np.random.default_rng(90)
    x = rng.random((400, 90))
    b = rng.random((90, 3)) * (rng.random((90, 3)) > 0.4)
    y = x @ b + 0.01 * rng.standard_normal((400, 3))
    b0 = rng.random((90, 3))
    r = nnlm_amd.c_nnlm(x, y, [0, 0, 0], None, b0, 2000, 1e-10, 1, 1)
    o = ref.c_nnlm(x, y, [0, 0, 0], None, b0, 2000, 1e-10, 1, 1)
    assert relF(r["coefficient"], o["coefficient"]) < 1e-9 and r["n_iteration"] == o["n_iteration"]
    assert np.all(r["coefficient"] >= 0)


@pytest.mark.parametrize("pname,prec,tol", [("f64", _lib.PREC_F64, 1e-10), ("f32", _lib.PREC_F32, 1e-4)])
@pytest.mark.parametrize("method", [3, 4])
@pytest.mark.parametrize("n", [23000, 40000, 41500])
def test_kl_contraction_longer_than_32768(pname, prec, tol, method, n):
    """nnmf(loss = 'mkl') on an n x 9 matrix: the H half-step contracts over n rows (the reference streams any length,
    src/base_algorithms.cpp:71-151), the W half-step solves n columns.  F32 mode: 23000 and 40000 take the one-row-buffer form of
    kl_tile_kernel (12 / 20 pieces per thread), 41500 kl_stream_kernel; strict mode: kl_stream_kernel beyond 20480."""
    rng = np.random.default_rng(method)
    m, k = 9, 3
    A = rng.random((n, m))
    A[rng.random((n, m)) < 0.02] = np.nan
    W0, H0 = rng.random((n, k)), rng.random((k, m))
    reg = [0.01, 0.0, 0.02]
    with nnlm_amd.Handle(0, prec) as h:
        h.set_matrix(A)
        h.set_factors(k, W0, H0)
        h.half_step(1, reg, 3, 1e-9, method)
        _, H1 = h.get_factors()
        sw = h.take_sweeps()
        H_ref, it = ref.update(H0, W0.T.copy(), A, None, reg, 3, 1e-9, method)
        assert relF(H1, H_ref) < tol
        if pname == "f64":
            assert sw == it
        h.half_step(0, reg, 2, 1e-9, method)
        W1, _ = h.get_factors()
        Wt_ref, _ = ref.update(W0.T.copy(), H_ref, A.T.copy(), None, reg, 2, 1e-9, method)
        assert relF(W1, Wt_ref.T) < 10 * tol


@pytest.mark.parametrize("pname,prec,tol", PRECS)
@pytest.mark.parametrize("k", [16, 17, 18, 32, 33, 40, 49, 50, 64])
def test_missing_values_per_column_gram_every_tile_form(pname, prec, tol, k):
    """update_with_missing (src/update_with_missing.cpp:58-139) over the forms of the per-column Gram kernel (k_missing.h,
    na_gram_lds_kernel): whole tiles (16, 32, 64), tail coordinates on the VALU (17, 18, 33, 49, 50), a padded last tile (40);
    columns with no missing entry (empty row list), with one, with 2-3 rows beyond a multiple of four, with more than half missing
    (the list then holds the PRESENT rows), and an all-missing column."""
    rng = np.random.default_rng(1000 + k)
    n, m = 203, 37
    A = rng.random((n, m)) + 0.1
    A[rng.random((n, m)) < 0.1] = np.nan
    A[:, 0] = rng.random(n) + 0.1          # no missing entry
    A[:, 1] = rng.random(n) + 0.1
    A[5, 1] = np.nan                        # exactly one
    A[:, 2] = rng.random(n) + 0.1
    A[[3, 9, 100, 150, 151, 152, 200], 2] = np.nan  # 7 = 4 + 3
    A[rng.random(n) < 0.7, 3] = np.nan     # mostly missing
    A[:, 4] = np.nan                        # nothing observed
    A[7, :] = np.nan                        # a row of the W half-step with nothing observed
    W0, H0 = rng.random((n, k)), rng.random((k, m))
    reg = [0.02, 0.01, 0.03]
    with nnlm_amd.Handle(0, prec) as h:
        h.set_matrix(A)
        h.set_factors(k, W0, H0)
        if pname == "f32":
            tol = 1e-4  # (round 6: the per-column solver runs the chain on fp32 state; cold half-step from random factors, see test_half_step_matches_oracle)
        h.half_step(1, reg, 4, 1e-9, 1)
        _, H1 = h.get_factors()
        s1 = h.take_sweeps()
        H_ref, it1 = ref.update(H0, W0.T.copy(), A, None, reg, 4, 1e-9, 1)
        assert relF(H1, H_ref) < tol
        h.half_step(0, reg, 4, 1e-9, 1)
        W1, _ = h.get_factors()
        s2 = h.take_sweeps()
        Wt_ref, it2 = ref.update(W0.T.copy(), H_ref, A.T.copy(), None, reg, 4, 1e-9, 1)
        assert relF(W1, Wt_ref.T) < 10 * tol
        if pname == "f64":
            assert (s1, s2) == (it1, it2)


@pytest.mark.parametrize("pname,prec,tol", PRECS)
def test_missing_values_row_lists_around_the_gather_steps(pname, prec, tol):
    """Row lists of exactly L rows for L around the multiples of the gather step (32 rows), of the ring depths (2 stage buffers, 4
    index slots) and of the fp32 -> fp64 fold (256 rows) of na_gram_f16_kernel / the four-row groups of na_gram_lds_kernel
    (k_missing.h); both sides of the half-way point, where the list switches from the missing to the present rows."""
    rng = np.random.default_rng(4242)
    lens = [0, 1, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 159, 160, 161, 255, 256, 257, 288, 511, 512, 513, 600, 649, 650, 651, 700, 1299]
    n, m, k = 1300, len(lens), 50
    A = rng.random((n, m)) + 0.1
    for j, L in enumerate(lens):
        A[rng.choice(n, L, replace=False), j] = np.nan
    W0, H0 = rng.random((n, k)), rng.random((k, m))
    reg = [0.02, 0.01, 0.03]
    if pname == "f32":
        tol = 1e-4  # (fp32 chain, cold half-step: see test_half_step_matches_oracle)
    with nnlm_amd.Handle(0, prec) as h:
        h.set_matrix(A)
        h.set_factors(k, W0, H0)
        h.half_step(1, reg, 4, 1e-9, 1)
        _, H1 = h.get_factors()
        s1 = h.take_sweeps()
        H_ref, it1 = ref.update(H0, W0.T.copy(), A, None, reg, 4, 1e-9, 1)
        for j in range(m):
            assert relF(H1[:, j], H_ref[:, j]) < 10 * tol, lens[j]
        assert relF(H1, H_ref) < tol
        if pname == "f64":
            assert s1 == it1


@pytest.mark.parametrize("seed", range(16))
def test_random_small_shapes_with_missing_values(seed):
    """Randomised edge sweep of the NA path (row lists shorter than one gather step, contraction lengths below one tile, ranks that
    use every instantiation of the Gram / solver kernels, 0 - 90 % missing, all-missing columns): both half-steps against the oracle
    in both modes."""
    rng = np.random.default_rng(9000 + seed)
    n, m = int(rng.integers(1, 150)), int(rng.integers(1, 150))
    k = int(rng.integers(1, 65))
    rate = [0.02, 0.1, 0.5, 0.9][seed % 4]
    method = 1 + (seed // 4) % 2
    A = rng.random((n, m)) + 0.05
    A[rng.random((n, m)) < rate] = np.nan
    if not np.isnan(A).any():
        A[0, 0] = np.nan
    W0, H0 = rng.random((n, k)), rng.random((k, m))
    reg = [0.05, 0.01, 0.02]
    for prec, tol in ((_lib.PREC_F64, 1e-9), (_lib.PREC_F32, 1e-4)):
        with nnlm_amd.Handle(0, prec) as h:
            h.set_matrix(A)
            h.set_factors(k, W0, H0)
            h.half_step(1, reg, 3, 1e-9, method)
            _, H1 = h.get_factors()
            H_ref, _ = ref.update(H0, W0.T.copy(), A, None, reg, 3, 1e-9, method)
            assert relF(H1, H_ref) < tol, (n, m, k, rate, method, prec)
            h.half_step(0, reg, 3, 1e-9, method)
            W1, _ = h.get_factors()
            Wt_ref, _ = ref.update(W0.T.copy(), H_ref, A.T.copy(), None, reg, 3, 1e-9, method)
            assert relF(W1, Wt_ref.T) < 10 * tol, (n, m, k, rate, method, prec)


# ---- column-sharded half-steps (missing values, KL methods): all-gather only -------------------------------------------
@pytest.mark.parametrize("pname,prec,tol", PRECS)
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["dense_scd", "dense_lee", "dense_scd_k50", "dense_scd_k80", "na_scd", "na_lee", "na_scd_k80", "kl_scd", "kl_lee",
                                  "kl_lee_k80", "na_kl_lee"])
def test_virtual_ranks_run_column_sharded_half_steps(pname, prec, tol, world, case):
    """The column-sharded form -- the default for dense square loss, the only one with missing values (per-column Grams) and for the
    KL methods: the column is the unit of the multi-GPU split, a rank does
    all the work of ITS columns (nnlm_shard_cols) over the whole contraction into a packed slab, ONE all-gather returns the
    factor, every rank unpacks -- no all-reduce.  `world` virtual ranks on one device, the host standing in for ncclAllGather;
    every rank must end with identical factors, equal to the single-rank result."""
    import re
    base, kk = re.fullmatch(r"(.*?)(?:_k(\d+))?", case).groups()
    method = {"dense_scd": 1, "dense_lee": 2, "na_scd": 1, "na_lee": 2, "kl_scd": 3, "kl_lee": 4, "na_kl_lee": 4}[base]
    rng = np.random.default_rng(world + method)
    # (k = 50: the tail-block sweep on column slabs that do not start at 0; 80: the rank > 64 kernels, k_generic.h)
    n, m, k = 700, 333, (int(kk) if kk else 13)
    A = rng.random((n, 5)) @ rng.random((5, m)) + 0.1 * rng.random((n, m))
    if case.startswith("na"):
        A.ravel()[rng.choice(A.size, A.size // 10, replace=False)] = np.nan
    W0, H0 = 0.3 * rng.random((n, k)), 0.3 * rng.random((k, m))
    Wm = rng.random((n, k)) < 0.05
    reg = [0.02, 0.01, 0.03]
    inner = 5 if method < 3 else 2
    if method >= 3 and pname == "f32":
        tol = 1e-4
    with nnlm_amd.Handle(0, prec) as h1:
        h1.set_matrix(A)
        h1.set_factors(k, W0, H0, Wm, None)
        h1.iterate(2, reg, reg, inner, 1e-9, method)
        W_ref, H_ref = h1.get_factors()
        sw_ref = h1.take_sweeps()
        mse_ref = h1.errors()[0]
    hs = [nnlm_amd.Handle(0, prec) for _ in range(world)]
    try:
        for rk, h in enumerate(hs):
            h.comm_init(None, rk, world)
            h.set_matrix(A)
            h.set_factors(k, W0, H0, Wm, None)
        for _ in range(2):
            for which in (0, 1):
                for phase in (1, 2):
                    for h in hs:
                        h.debug_phase(which, phase, reg, inner, 1e-9, method)
                _lib.debug_exchange(hs, which, 2)  # the all-gather (phase 1 left nothing to all-reduce)
                for h in hs:
                    h.debug_phase(which, 3, reg, inner, 1e-9, method)
        res = [h.get_factors() for h in hs]
        sweeps = sum(h.take_sweeps() for h in hs)
        mse = sum(h.errors()[0] for h in hs)
    finally:
        for h in hs:
            h.close()
    for W, H in res[1:]:
        assert np.array_equal(W, res[0][0]) and np.array_equal(H, res[0][1])
    # the same kernels on the same columns: only the split-K depth of the cross product differs between the sharded and the
    # single-rank launch (fewer column tiles per launch), i.e. the order of a few fp64 additions
    t = 1e-11 if pname == "f64" else tol
    assert relF(res[0][0], W_ref) < t and relF(res[0][1], H_ref) < t
    assert np.array_equal(res[0][0][Wm], W0[Wm])
    assert abs(sweeps - sw_ref) <= (0 if pname == "f64" else 2)
    assert abs(mse - mse_ref) < (1e-9 if pname == "f64" else 1e-5) * mse_ref


@pytest.mark.parametrize("case", ["na_scd", "kl_lee"])
def test_column_sharded_path_with_a_real_one_rank_rccl_communicator(case):
    """The same code path end to end with a real RCCL communicator of size 1 (ncclAllGather of the packed slab, unpack)."""
    method = 1 if case == "na_scd" else 4
    rng = np.random.default_rng(5)
    n, m, k = 300, 200, 9
    A = rng.random((n, m))
    if case == "na_scd":
        A.ravel()[rng.choice(A.size, A.size // 10, replace=False)] = np.nan
    W0, H0 = rng.random((n, k)), rng.random((k, m))
    reg = [0.02, 0.01, 0.03]
    out = []
    for sharded in (False, True):
        with nnlm_amd.Handle(0, _lib.PREC_F64) as h:
            if sharded:
                h.comm_init(_lib.comm_unique_id(), 0, 1)
            h.set_matrix(A)
            h.set_factors(k, W0, H0)
            r = h.run(reg, reg, 4, -1.0, 0, False, 3, 1e-9, method, 2)
            W, H = h.get_factors()
            out.append((W, H, r))
    a, b = out
    assert relF(a[0], b[0]) < 1e-12 and relF(a[1], b[1]) < 1e-12
    assert np.allclose(a[2]["mse_error"], b[2]["mse_error"], rtol=1e-12) and np.array_equal(a[2]["average_epoch"], b[2]["average_epoch"])
