"""Host logic of the R-interface mirror (nnlm_amd/api.py) and the C-ABI surface, on CPU.

The compute engine here is the oracle, composed by the TEST around prepare_*/finish_* (the product's
nnmf()/nnlm() are hard-wired to the HIP library and must fail loudly without it)."""
import ctypes
import os
import re
import sys
import warnings

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import kat_case1, kat_case2, r_all_equal  # noqa: E402
import nnlm_amd  # noqa: E402
from nnlm_amd import _lib, api  # noqa: E402
from oracle import ref  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_nnmf(*a, **kw):
    args, ctx = api.prepare_nnmf(*a, **kw)
    return api.finish_nnmf(ref.c_nnmf(*args), ctx)


def oracle_nnlm(*a, **kw):
    kw.pop("rng", None)
    args, ctx = api.prepare_nnlm(*a, **kw)
    args = list(args)
    if args[4] is None:  # beta0 empty -> the reference draws U(0,1); any positive start reaches the same optimum
        args[4] = np.full((args[0].shape[1], args[1].shape[1]), 0.5)
    return api.finish_nnlm(ref.c_nnlm(*args), ctx)


# ---- C ABI surface -------------------------------------------------------------------------------
def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "nnlm_mi355x.h")).read()
    declared = set(re.findall(r"\b(nnlm_[a-z0-9_]+)\s*\(", header)) - {"nnlm_callbacks"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.nnlm_abi_version() == 1
    assert lib.nnlm_trace_capacity(500, 2) == 251 and lib.nnlm_trace_capacity(5, 0) == 6


def test_product_path_fails_loudly_without_gpu(gpu_available):
    if gpu_available:
        pytest.skip("GPU present")
    with pytest.raises(nnlm_amd.NnlmError, match="no HIP device"):
        nnlm_amd.Handle()
    A = np.random.default_rng(0).random((20, 10))
    with pytest.raises(nnlm_amd.NnlmError):
        api.nnmf(A, 2)
    with pytest.raises(nnlm_amd.NnlmError):
        api.nnlm(A, A[:, 0])


def test_product_package_never_imports_the_oracle():
    pat = re.compile(r"(^|\n)\s*(from|import)\s+oracle\b|oracle/|libnnlm_ref|nnlm_ref\.c|CDLL\([^)]*ref")
    for root, _, files in os.walk(os.path.join(ROOT, "nnlm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c")):
                src = open(os.path.join(root, f)).read()
                assert not pat.search(src), f
                for target in re.findall(r'dlopen\("([^"]+)"', src):  # the only library loaded at run time is RCCL
                    assert "rccl" in target, (f, target)


# ---- R/misc.R ------------------------------------------------------------------------------------
def test_method_codes():
    assert [api.get_method_code(m, l) for m, l in (("scd", "mse"), ("lee", "mse"), ("scd", "mkl"), ("lee", "mkl"))] == [1, 2, 3, 4]
    with pytest.raises(api.NnlmStop):
        api.get_method_code("foo", "mse")


def test_reformat_input_stacking_order_and_masks():
    n, m, k = 6, 4, 2
    rng = np.random.default_rng(0)
    W0, H0 = rng.random((n, 3)), rng.random((1, m))
    im = api.reformat_input({"W0": W0, "H0": H0}, None, n, m, k, rng=np.random.default_rng(1))
    assert im["K"] == 6 and im["Wi"].shape == (n, 6) and im["Hi"].shape == (6, m)
    # [W W0 W1] / [H; H1; H0]: known W profiles in columns k..k+2, known H profile in the LAST row
    assert np.array_equal(im["Wi"][:, 2:5], W0) and np.array_equal(im["Hi"][5:6, :], H0)
    assert im["Wm"][:, 2:5].all() and not im["Wm"][:, :2].any() and not im["Wm"][:, 5:].any()
    assert im["Hm"][5:, :].all() and not im["Hm"][:5, :].any()
    # nothing supplied: empty blocks -> the engine draws the default init (src/nnmf.cpp:82-98)
    im = api.reformat_input(None, None, n, m, k)
    assert im["Wi"].shape == (n, 0) and im["Hi"].shape == (0, m) and im["Wm"].shape == (n, 0) and im["K"] == k
    # mask only on H
    Hm = rng.random((k, m)) < 0.5
    im = api.reformat_input(None, {"H": Hm}, n, m, k)
    assert im["Hm"].shape == (k, m) and np.array_equal(im["Hm"], Hm) and im["Wm"].shape == (n, 0)
    with pytest.raises(api.NnlmStop, match="Dimension of matrix mask\\$W"):
        api.reformat_input(None, {"W": np.zeros((n + 1, k), dtype=bool)}, n, m, k)
    with pytest.raises(api.NnlmStop, match="must be logical"):
        api.reformat_input(None, {"W": np.zeros((n, k))}, n, m, k)


def test_prepare_nnmf_defaults_match_r():
    A = np.random.default_rng(0).random((30, 20))
    args, ctx = api.prepare_nnmf(A, 3)
    assert args[8:] == (500, 1e-4, 1, 1, True, 50, 1e-9, 1, 2)  # max.iter, rel.tol, n.threads, verbose, warn, inner, tol, code, trace
    args, _ = api.prepare_nnmf(A, 3, loss="mkl", method="lee")
    assert args[13] == 1 and args[15] == 4 and args[16] == 100
    args, _ = api.prepare_nnmf(A, 3, inner_max_iter=30)
    assert args[16] == 3  # as.integer(100/30)
    args, _ = api.prepare_nnmf(A, 3, trace=0)
    assert args[16] == 999999
    args, _ = api.prepare_nnmf(A, 3, alpha=0.1, beta=[0.01, 0.02])
    assert list(args[6]) == [0.1, 0, 0] and list(args[7]) == [0.01, 0.02, 0]


def test_check_k_rule():
    A = np.random.default_rng(0).random((50, 10))
    with pytest.raises(api.NnlmStop, match="k larger than 10 is not recommended"):
        api.prepare_nnmf(A, 20)  # test-nnmf.R:60
    api.prepare_nnmf(A, 20, check_k=False)
    api.prepare_nnmf(A, 20, alpha=0.1)
    A2 = A.copy()
    A2[:45, 0] = np.nan  # column 0 keeps 5 observations
    with pytest.raises(api.NnlmStop, match="k larger than 5"):
        api.prepare_nnmf(A2, 6)


def test_nnmf_wrapper_roundtrip_warning_and_wnorm():
    rng = np.random.default_rng(234)
    A = rng.random((50, 3)) @ rng.random((3, 10))
    r = oracle_nnmf(A, 3, max_iter=10000, rel_tol=1e-8, init={"W": 0.01 * rng.random((50, 3)), "H": 0.01 * rng.random((3, 10))})
    assert r_all_equal(r.W @ r.H, A) and r.options["method"] == "scd" and r.options["trace"] == 2
    with pytest.warns(RuntimeWarning, match="Target tolerance not reached. Try a larger max.iter."):  # test-nnmf.R:57-58
        oracle_nnmf(A, 2, alpha=0.1, beta=0, max_iter=10, init={"W": rng.random((50, 2)), "H": rng.random((2, 10))})
    r2 = oracle_nnmf(A, 3, max_iter=50, W_norm=1, init={"W": rng.random((50, 3)), "H": rng.random((3, 10))}, show_warning=False)
    assert np.allclose(r2.W.sum(axis=0), 1.0)
    r3 = oracle_nnmf(A, 3, max_iter=50, W_norm=np.inf, init={"W": rng.random((50, 3)), "H": rng.random((3, 10))}, show_warning=False)
    assert np.allclose(r3.W.max(axis=0), 1.0)
    assert "Non-negative matrix factorization" in repr(r)


def test_known_profiles_stay_fixed():
    """test-nnmf.R:41-47: init=list(W0=, H0=) columns/rows are never updated."""
    rng = np.random.default_rng(2)
    n, m, k = 50, 10, 3
    A = rng.random((n, k)) @ rng.random((k, m))
    W1, H2 = rng.random((n, 2)), np.ones((1, m))
    A2 = A + W1 @ rng.random((2, m)) + rng.random((n, 1)) @ H2
    r = oracle_nnmf(A2, k, init={"W0": W1, "H0": H2}, max_iter=1000, rel_tol=1e-3, inner_max_iter=20, rng=np.random.default_rng(0))
    assert r.W.shape == (n, 6) and r.H.shape == (6, m)
    assert np.array_equal(r.W[:, 3:5], W1) and np.array_equal(r.H[5:, :], H2)


# ---- nnlm / predict -------------------------------------------------------------------------------
def test_nnlm_kats_through_the_wrapper():
    A, b, _ = kat_case1()
    sol = oracle_nnlm(A, A @ b)
    assert sol.coefficients.shape == (5,) and r_all_equal(sol.coefficients, b)  # test-nnlm.R:14-15
    A, b2, _ = kat_case2()
    sol2 = oracle_nnlm(A, A @ b2)
    assert sol2.coefficients.shape == (5, 2) and r_all_equal(sol2.coefficients, b2)
    assert set(sol2.error) == {"MSE", "MKL", "target.error"} and sol2.error["MSE"] < 1e-20


def test_nnlm_errors_and_warnings():
    rng = np.random.default_rng(123)
    A = rng.random((5, 4))
    with pytest.raises(api.NnlmStop, match="Dimensions of x and y do not match."):  # test-nnlm.R:54
        api.prepare_nnlm(A, rng.random(4))
    with pytest.warns(RuntimeWarning, match="x does not have a full column rank. Solution may not be unique."):  # :61
        api.prepare_nnlm(A.T, np.arange(1.0, 5.0)[:, None])
    with pytest.raises(api.NnlmStop, match="max.iter must be positive."):
        api.prepare_nnlm(A, rng.random(5), max_iter=0)
    with pytest.raises(api.NnlmStop, match="contains missing values"):
        B = A.copy()
        B[2, 1] = np.nan
        api.prepare_nnlm(B, rng.random(5))
    with pytest.warns(RuntimeWarning, match="negative values"):
        api.prepare_nnlm(A, -rng.random(5), loss="mkl", check_x=False)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        args, _ = api.prepare_nnlm(A, rng.random(5), mask=np.array([[True], [False], [False], [False]]))
    assert np.array_equal(args[4].ravel(), [0.0, 1.0, 1.0, 1.0])  # masked & uninitialised -> fixed to 0, R/nnlm.R:110-112


def test_predict_nnmf_shapes():
    rng = np.random.default_rng(0)
    A = rng.random((50, 10))
    r = oracle_nnmf(A, 2, alpha=0.1, beta=0.01, init={"W": rng.random((50, 2)), "H": rng.random((2, 10))}, show_warning=False)
    assert np.allclose(api.predict_nnmf(r, which="A"), r.W @ r.H)
    Wn = api.predict_nnmf(r, A[:4, :], which="W", _nnlm=oracle_nnlm)  # test-nnmf.R:55
    assert Wn["coefficients"].shape == (4, 2)
    Hn = api.predict_nnmf(r, A[:, :3], which="H", _nnlm=oracle_nnlm)
    assert Hn["coefficients"].shape == (2, 3)
    with pytest.raises(api.NnlmStop):
        api.predict_nnmf(r, A[:4, :5], which="W", _nnlm=oracle_nnlm)


def test_mse_mkl():
    obs = np.array([1.0, 2.0, np.nan, 4.0])
    pred = np.array([1.5, 2.0, 3.0, 3.0])
    e = api.mse_mkl(obs, pred)
    assert np.isclose(e["MSE"], (0.25 + 0 + 1) / 3)
    assert np.isnan(api.mse_mkl(np.array([-1.0, 2.0]), np.array([1.0, 2.0]), show_warning=False)["MKL"])


def test_r_glue_compiles_against_the_r_api_surface_it_uses(tmp_path):
    """pkg/src/r_glue.c (the .Call stubs of the drop-in, reference src/RcppExports.cpp:10-65) must compile cleanly as C99 against
    the R API.  This image has no R: tests/r_stub/ declares exactly the entries the glue uses (test-only).  The (DL_FUNC)
    casts of the registration table are R's own idiom, hence -Wno-cast-function-type."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["gcc", "-fsyntax-only", "-std=c99", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type", "-I" + os.path.join(root, "tests", "r_stub"),
           "-I" + os.path.join(root, "include"), os.path.join(root, "pkg", "src", "r_glue.c")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the registration table carries the reference's two routines with the reference's arities (src/RcppExports.cpp:56-60)
    src = open(os.path.join(root, "pkg", "src", "r_glue.c")).read()
    assert '{"_NNLM_c_nnlm", (DL_FUNC)&_NNLM_c_nnlm, 9}' in src and '{"_NNLM_c_nnmf", (DL_FUNC)&_NNLM_c_nnmf, 17}' in src
    for f in ("DESCRIPTION", "NAMESPACE", os.path.join("src", "Makevars"), os.path.join("R", "calls.R")):
        assert os.path.exists(os.path.join(root, "pkg", f)), f
    ns = open(os.path.join(root, "pkg", "NAMESPACE")).read()
    assert "useDynLib(NNLM, .registration = TRUE)" in ns and "import(Rcpp)" not in ns
