"""Randomised whole-driver parity: nnlm_amd.c_nnmf (the C ABI of libnnlm_mi355x.so) against the oracle's ref.c_nnmf
(oracle/nnlm_ref.c, the restatement of src/nnmf.cpp:4-219) over random shapes, ranks, the four methods, missing values, masks,
regularisation, trace strides and inner iteration limits -- the combinations the hand-written cases of test_gpu_parity.py do not
enumerate.  NNLM_FUZZ_SEEDS (default 16) sets the number of cases per test; the round's deep run used 300 (900 cases with test_random_nnlm_runs: 863 agree, 37 degenerate;
120 seeds of the virtual-rank runs and 150 of the stopping rule: all agree, 5 degenerate; a later run of
700 seeds of the first four tests: 2714 agree, 85 degenerate, 1 sweep count off by 2 on a rank-deficient per-column Gram).

Strict mode: factors at 1e-9 relative Frobenius, iteration counts, trace lengths and sweep counts (average_epoch) exact.
F32 mode: north_star's 1e-4 on well-conditioned cases (rank at most a third of the smaller dimension, at most 30 % missing)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import relF  # noqa: E402
import nnlm_amd  # noqa: E402
from nnlm_amd import _lib  # noqa: E402
from oracle import ref  # noqa: E402

pytestmark = pytest.mark.gpu

SEEDS = int(os.environ.get("NNLM_FUZZ_SEEDS", "16"))


def make_case(seed, well_conditioned):
    rng = np.random.default_rng(77000 + seed)
    n, m = int(rng.integers(2, 400)), int(rng.integers(2, 400))
    if seed % 7 == 0:
        n = int(rng.integers(400, 1500))  # more than one cross-product tile / gather step
    if seed % 11 == 0:
        n, m = int(rng.integers(1500, 4000)), int(rng.integers(400, 2500))  # several tiles, split-K slabs and sweep workgroups each way
    kmax = max(1, min(n, m) // 3) if well_conditioned else min(n, m, 70)
    k = int(rng.integers(1, min(kmax, 64 if well_conditioned or seed % 5 else 70) + 1))
    method = 1 + seed % 4
    # A planted non-negative model of rank k + 3 plus noise, and a start near its factors.  With structureless data or a random start
    # whole factors die on the way (a column of W exactly zero: the Gram diagonal of the next half-step is NNLM_TINY and the
    # coordinate's value, mu / 1e-16, is decided by the rounding of the Gram), and the Newton steps of the KL coordinate descent
    # empty whole rows (y-hat = 0: mu = w / (0 + 1e-16), src/base_algorithms.cpp:90, turns 1e-19 of rounding dust into 1e-3) -- there
    # no two fp64 implementations agree, the reference at two summation orders included; degenerate() skips what still gets there.
    Wp, Hp = rng.random((n, k + 3)) ** 2 + 0.05, rng.random((k + 3, m)) ** 2 + 0.05
    A = Wp @ Hp / (k + 3) * 4 + 0.02 * rng.random((n, m)) + 0.01
    na = [0.0, 0.0, 0.05, 0.3][(seed // 4) % 4] if well_conditioned else [0.0, 0.1, 0.6, 0.9][(seed // 4) % 4]
    if na > 0:
        A[rng.random((n, m)) < na] = np.nan
    Wm = Hm = None
    if (seed // 3) % 3 == 1:  # masks: a few coordinates pinned to zero (R/nnmf.R:182-196)
        Wm = (rng.random((n, k)) < 0.1).astype(np.int32)
        Hm = (rng.random((k, m)) < 0.1).astype(np.int32)
    sc = 2.0 / np.sqrt(k + 3)
    W0, H0 = Wp[:, :k] * sc * (0.7 + 0.6 * rng.random((n, k))), Hp[:k, :] * sc * (0.7 + 0.6 * rng.random((k, m)))
    if Wm is not None:
        W0[Wm != 0] = 0.0
        H0[Hm != 0] = 0.0
    reg_choices = ([0, 0, 0], [0.01, 0, 0.01], [0.02, 0.01, 0.03], [0, 0.05, 0])
    alpha, beta = list(reg_choices[seed % 4]), list(reg_choices[(seed // 2) % 4])
    max_iter = int(rng.integers(1, 6))
    trace = int(rng.integers(1, 4))
    inner = int(rng.integers(1, 8)) if method < 3 else int(rng.integers(1, 4))
    return dict(A=A, k=k, W0=W0, H0=H0, Wm=Wm, Hm=Hm, alpha=alpha, beta=beta, max_iter=max_iter, trace=trace, inner=inner, method=method)


def nnmf_args(c, max_iter):
    return (c["A"], c["k"], c["W0"], c["H0"], c["Wm"], c["Hm"], c["alpha"], c["beta"], max_iter, -1.0, 1, 0, False, c["inner"], 1e-9,
            c["method"], c["trace"])


def run_both(c):
    return nnlm_amd.c_nnmf(*nnmf_args(c, c["max_iter"])), ref.c_nnmf(*nnmf_args(c, c["max_iter"]))


def degenerate(c):
    """True if, on the oracle's way, a factor dies (a column of W / row of H at 1e-8 of the median norm) or a row of W H vanishes
    where A is observed (see make_case): such a run is not reproducible beyond its error traces."""
    obs = np.isfinite(c["A"])
    for it in range(1, c["max_iter"] + 1):
        o = ref.c_nnmf(*nnmf_args(c, it))
        W, H = o["W"], o["H"]
        if not (np.isfinite(W).all() and np.isfinite(H).all()):
            return True
        dw, dh = (W * W).sum(axis=0), (H * H).sum(axis=1)
        if dw.min() < 1e-8 * np.median(dw) or dh.min() < 1e-8 * np.median(dh):
            return True
        if c["method"] >= 3 and ((W @ H)[obs] < 1e-9).any():
            return True
    return False


def describe(c):
    return dict(shape=c["A"].shape, k=c["k"], method=c["method"], na=float(np.isnan(c["A"]).mean()), masks=c["Wm"] is not None, alpha=c["alpha"],
                beta=c["beta"], max_iter=c["max_iter"], trace=c["trace"], inner=c["inner"])


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_driver_runs_strict_mode(monkeypatch, seed):
    monkeypatch.setenv("NNLM_PRECISION", "f64")
    c = make_case(seed, well_conditioned=False)
    if degenerate(c):
        pytest.skip("a factor dies on the way: not reproducible (see make_case)")
    r, o = run_both(c)
    d = describe(c)
    assert r["n_iteration"] == o["n_iteration"], d
    for key in ("mse_error", "mkl_error", "target_error", "average_epoch"):
        assert r[key].shape == o[key].shape, (key, d)
    obs = np.isfinite(c["A"])
    if min(obs.sum(axis=0).min(), obs.sum(axis=1).min()) >= c["k"]:
        assert np.array_equal(r["average_epoch"], o["average_epoch"]), d
    else:
        # a column with fewer observed entries than coordinates: its Gram is rank deficient, the coordinates of its null space sit on
        # NNLM_TINY and whether their last 1e-17 counts as a change is decided by the rounding of the Gram (1 case in 700 seeds)
        n, m = c["A"].shape
        assert np.allclose(r["average_epoch"], o["average_epoch"], rtol=0, atol=(2.0 * c["inner"] + 1e-9) / (n + m)), d
    assert relF(r["W"], o["W"]) < 1e-9 and relF(r["H"], o["H"]) < 1e-9, d
    assert np.allclose(r["mse_error"], o["mse_error"], rtol=1e-8, atol=1e-13), d
    assert np.allclose(r["mkl_error"], o["mkl_error"], rtol=1e-8, atol=1e-11), d
    assert np.allclose(r["target_error"], o["target_error"], rtol=1e-8, atol=1e-11), d


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_driver_runs_f32_mode(monkeypatch, seed):
    monkeypatch.setenv("NNLM_PRECISION", "f32")
    c = make_case(seed, well_conditioned=True)
    if degenerate(c):
        pytest.skip("a factor dies on the way: not reproducible (see make_case)")
    r, o = run_both(c)
    d = describe(c)
    assert r["n_iteration"] == o["n_iteration"], d
    assert r["mse_error"].shape == o["mse_error"].shape, d
    assert relF(r["W"], o["W"]) < 1e-4 and relF(r["H"], o["H"]) < 1e-4, (relF(r["W"], o["W"]), relF(r["H"], o["H"]), d)
    assert np.allclose(r["mse_error"], o["mse_error"], rtol=1e-3, atol=1e-12), d
    assert np.allclose(r["target_error"], o["target_error"], rtol=1e-3, atol=1e-9), d


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_nnlm_runs(monkeypatch, seed):
    """c_nnlm (src/nnlm.cpp:4-53: the half-step solver on y ~ x beta, rank = ncol(x)) with random predictors, responses with and
    without missing values, masks, given / default starts and the four methods; strict mode, sweep counts exact."""
    monkeypatch.setenv("NNLM_PRECISION", "f64")
    rng = np.random.default_rng(55000 + seed)
    n, p, q = int(rng.integers(3, 600)), int(rng.integers(1, 80)), int(rng.integers(1, 40))
    if seed % 3 == 0:
        p = int(rng.integers(1, min(n, 20) + 1))  # (a well-determined regression)
    method = 1 + seed % 4
    x = rng.random((n, p)) + 0.02
    b = rng.random((p, q)) * (rng.random((p, q)) > 0.3)
    b[rng.integers(0, p, q), np.arange(q)] += 0.3  # (no response without a predictor: see the skip below)
    y = x @ b + 0.02 * rng.random((n, q)) + 0.01
    if (seed // 4) % 3 == 1:
        y[rng.random((n, q)) < [0.05, 0.3, 0.7][seed % 3]] = np.nan
    mask = (rng.random((p, q)) < 0.15) if (seed // 2) % 3 == 1 else None
    b0 = None if seed % 5 == 0 else rng.random((p, q)) + 0.01
    if mask is not None and b0 is not None:
        b0[mask] = 0.0
    alpha = [[0, 0, 0], [0.01, 0, 0.001], [0.02, 0.01, 0.03]][seed % 3]
    max_iter = int(rng.integers(1, 30)) if method in (1, 2) else int(rng.integers(1, 6))
    r = nnlm_amd.c_nnlm(x, y, alpha, mask, b0, max_iter, 1e-10, 1, method)
    o = ref.c_nnlm(x, y, alpha, mask, b0, max_iter, 1e-10, 1, method)
    d = dict(n=n, p=p, q=q, method=method, na=float(np.isnan(y).mean()), mask=mask is not None, b0=b0 is not None, alpha=alpha, max_iter=max_iter)
    if not np.isfinite(o["coefficient"]).all():
        pytest.skip("the oracle's own result is not finite")
    free = np.ones((p, q), bool) if mask is None else ~mask
    if method == 3 and ((o["coefficient"] == 0) | ~free).all(axis=0).any():
        # every free coordinate of a column at exactly zero under the KL coordinate descent: its state vector is cancellation noise
        # of either sign and the next sweep's quotients w / (noise + 1e-16) are decided by the summation order
        pytest.skip("a column of the KL coordinate descent went to zero: not reproducible")
    assert relF(r["coefficient"], o["coefficient"]) < 1e-9, (relF(r["coefficient"], o["coefficient"]), d)
    assert r["n_iteration"] == o["n_iteration"], (r["n_iteration"], o["n_iteration"], d)
    if mask is not None and b0 is not None:
        assert np.all(r["coefficient"][mask] == 0)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_virtual_rank_runs(monkeypatch, seed):
    """The multi-GPU arithmetic with 2 - 8 virtual ranks on one device (the host standing in for ncclAllReduce / ncclAllGather,
    nnlm_debug_phase / nnlm_debug_exchange) over random shapes -- ranks with few or no columns included --, methods, missing values,
    masks and both forms of the dense half-step: two iterations, every rank must end with identical factors, equal to the
    single-rank run up to the summation order of the split."""
    rng = np.random.default_rng(91000 + seed)
    world = [2, 3, 4, 8][seed % 4]
    n, m = int(rng.integers(8, 1400)), int(rng.integers(8, 900))
    k = int(rng.integers(1, min(n, m, 64) + 1)) if seed % 9 else int(rng.integers(65, 72))
    k = min(k, n, m)
    method = 1 + (seed // 4) % 4
    na = (seed // 2) % 3 == 1
    reduce_form = (seed % 2 == 1) and method <= 2 and not na
    Wp, Hp = rng.random((n, k + 2)) ** 2 + 0.05, rng.random((k + 2, m)) ** 2 + 0.05
    A = Wp @ Hp / (k + 2) * 4 + 0.02 * rng.random((n, m)) + 0.01
    if na:
        A[rng.random((n, m)) < 0.15] = np.nan
    sc = 2.0 / np.sqrt(k + 2)
    W0, H0 = Wp[:, :k] * sc * (0.7 + 0.6 * rng.random((n, k))), Hp[:k, :] * sc * (0.7 + 0.6 * rng.random((k, m)))
    Wm = (rng.random((n, k)) < 0.05) if seed % 3 == 0 else None
    Hm = (rng.random((k, m)) < 0.05) if seed % 6 == 0 else None
    if Wm is not None:
        W0[Wm] = 0.0
    if Hm is not None:
        H0[Hm] = 0.0
    alpha, beta = [[0, 0, 0], [0.02, 0.01, 0.03]][seed % 2], [[0.01, 0, 0.01], [0, 0, 0]][(seed // 2) % 2]
    inner = int(rng.integers(1, 6)) if method < 3 else int(rng.integers(1, 3))
    d = dict(world=world, shape=(n, m), k=k, method=method, na=na, reduce=reduce_form, masks=(Wm is not None, Hm is not None), inner=inner)
    for pname, prec in (("f64", _lib.PREC_F64), ("f32", _lib.PREC_F32)):
        with nnlm_amd.Handle(0, prec) as h1:
            h1.set_matrix(A)
            h1.set_factors(k, W0, H0, Wm, Hm)
            h1.iterate(2, alpha, beta, inner, 1e-9, method)
            W_ref, H_ref = h1.get_factors()
            sw_ref = h1.take_sweeps()
            mse_ref = h1.errors()[0]
        hs = [nnlm_amd.Handle(0, prec) for _ in range(world)]
        try:
            for rk, h in enumerate(hs):
                h.comm_init(None, rk, world, form="reduce" if reduce_form else "cols")
                h.set_matrix(A)
                h.set_factors(k, W0, H0, Wm, Hm)
            for _ in range(2):
                for which, reg in ((0, alpha), (1, beta)):
                    for h in hs:
                        h.debug_phase(which, 1, reg, inner, 1e-9, method)
                    if reduce_form:
                        _lib.debug_exchange(hs, which, 1)
                    for h in hs:
                        h.debug_phase(which, 2, reg, inner, 1e-9, method)
                    _lib.debug_exchange(hs, which, 2)
                    for h in hs:
                        h.debug_phase(which, 3, reg, inner, 1e-9, method)
            res = [h.get_factors() for h in hs]
            sweeps = sum(h.take_sweeps() for h in hs)
            mse = sum(h.errors()[0] for h in hs)
        finally:
            for h in hs:
                h.close()
        for W, H in res[1:]:
            assert np.array_equal(W, res[0][0]) and np.array_equal(H, res[0][1]), (pname, d)
        t = 1e-10 if pname == "f64" else (2e-5 if method < 3 else 1e-4)
        assert relF(res[0][0], W_ref) < t and relF(res[0][1], H_ref) < t, (pname, relF(res[0][0], W_ref), relF(res[0][1], H_ref), d)
        assert abs(sweeps - sw_ref) <= (0 if pname == "f64" else 2 + sw_ref // 1000), (pname, sweeps, sw_ref, d)
        assert abs(mse - mse_ref) < (1e-9 if pname == "f64" else 1e-5) * mse_ref, (pname, mse, mse_ref, d)


@pytest.mark.parametrize("seed", range(max(1, SEEDS // 2)))  # (up to 80 iterations each: the slow one)
def test_random_driver_stopping_rule(monkeypatch, seed):
    """The stopping rule of the driver (src/nnmf.cpp:142-158: relative change of the target error between two trace points below
    rel.tol) on random problems: the strict mode must stop at the reference's iteration with the reference's traces; the F32 mode
    may differ by one trace interval when the decisive quotient sits within its 1e-4 of the threshold."""
    c = make_case(seed, well_conditioned=True)
    c["max_iter"] = 80
    rel_tol = [1e-2, 1e-3, 1e-4][seed % 3]
    if degenerate(dict(c, max_iter=3)):
        pytest.skip("a factor dies on the way: not reproducible (see make_case)")
    args = (c["A"], c["k"], c["W0"], c["H0"], c["Wm"], c["Hm"], c["alpha"], c["beta"], c["max_iter"], rel_tol, 1, 0, False, c["inner"], 1e-9,
            c["method"], c["trace"])
    o = ref.c_nnmf(*args)
    d = dict(describe(c), rel_tol=rel_tol, n_iteration=o["n_iteration"])
    monkeypatch.setenv("NNLM_PRECISION", "f64")
    r = nnlm_amd.c_nnmf(*args)
    assert r["n_iteration"] == o["n_iteration"] and r["warning"] == o["warning"], (r["n_iteration"], d)
    assert r["target_error"].shape == o["target_error"].shape and np.allclose(r["target_error"], o["target_error"], rtol=1e-7, atol=1e-12), d
    assert np.array_equal(r["average_epoch"], o["average_epoch"]), d
    assert relF(r["W"], o["W"]) < 1e-8 and relF(r["H"], o["H"]) < 1e-8, (relF(r["W"], o["W"]), relF(r["H"], o["H"]), d)
    monkeypatch.setenv("NNLM_PRECISION", "f32")
    r = nnlm_amd.c_nnmf(*args)
    assert abs(r["n_iteration"] - o["n_iteration"]) <= c["trace"], (r["n_iteration"], d)
    if r["n_iteration"] == o["n_iteration"]:
        assert np.allclose(r["target_error"], o["target_error"], rtol=1e-3, atol=1e-9), d
