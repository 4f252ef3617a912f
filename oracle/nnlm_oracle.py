"""TEST INFRASTRUCTURE ONLY -- fp64 numpy restatement of linxihui/NNLM's nnmf()/nnlm() hot path.

This file is the *oracle*: a CPU restatement of the reference algorithm used by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` to
check the HIP path.  Nothing under ``nnlm_amd/`` imports it; the product path never
routes through it.

Parity pin: the restatement reproduces the reference's own known-answer vectors
(``tests/testthat/test-nnlm.R:6-15,19-26,29-43``; fixtures in ``tests/golden/nnlm_kat.json``)
and satisfies the properties of ``tests/testthat/test-nnmf.R``.  The reference itself
(R + Rcpp + RcppArmadillo + RcppProgress + R's BLAS) is unbuildable in this image, so
bit-level floating-point parity (BLAS summation order) is *unpinned*; value-level parity is
pinned to the reference's test tolerances.

All arrays are fp64.  Matrices follow the reference's in-C++ storage:
``Wt`` is k x n (W transposed, src/nnmf.cpp:90), ``H`` is k x m, ``A`` is n x m.
Citations are relative to /root/reference.
"""
from __future__ import annotations

import math
import numpy as np

TINY_NUM = 1e-16  # src/nnlm.h:17


# --------------------------------------------------------------------------------------
# per-column solvers (src/base_algorithms.cpp)
# --------------------------------------------------------------------------------------
def scd_ls_update(Hj, WtW, mu, mask, max_iter, rel_tol):
    """src/base_algorithms.cpp:3-37.  Hj (k,), mu (k,) are updated in place; returns sweeps."""
    k = WtW.shape[1]
    rel_err = 1.0 + rel_tol
    is_masked = mask is not None and mask.size > 0
    t = 0
    while t < max_iter and rel_err > rel_tol:
        rel_err = 0.0
        for q in range(k):
            if is_masked and mask[q] != 0:  # `mask(k) > 0` on an unsigned matrix
                continue
            tmp = Hj[q] - mu[q] / WtW[q, q]
            if tmp < 0:
                tmp = 0.0
            if tmp != Hj[q]:
                mu += (tmp - Hj[q]) * WtW[:, q]
            else:
                continue
            etmp = 2 * abs(Hj[q] - tmp) / (tmp + Hj[q] + TINY_NUM)
            if etmp > rel_err:
                rel_err = etmp
            Hj[q] = tmp
        t += 1
    return t


def lee_ls_update(Hj, WtW, WtAj, beta3, mask, max_iter, rel_tol):
    """src/base_algorithms.cpp:40-68."""
    k = WtW.shape[1]
    rel_err = rel_tol + 1.0
    is_masked = mask is not None and mask.size > 0
    t = 0
    while t < max_iter and rel_err > rel_tol:
        rel_err = 0.0
        for q in range(k):
            if is_masked and mask[q] != 0:  # `mask(k) > 0` on an unsigned matrix
                continue
            tmp = float(np.dot(WtW[:, q], Hj)) + beta3
            tmp = WtAj[q] / (tmp + TINY_NUM)
            Hj[q] *= tmp
            tmp = 2 * abs(tmp - 1) / (tmp + 1)
            if tmp > rel_err:
                rel_err = tmp
        t += 1
    return t


def scd_kl_update(Hj, Wt, Aj, sumW, mask, beta, max_iter, rel_tol):
    """src/base_algorithms.cpp:71-116."""
    k = Wt.shape[0]
    sumHj = float(np.sum(Hj))
    Ajt = Wt.T @ Hj
    rel_err = 1.0 + rel_tol
    is_masked = mask is not None and mask.size > 0
    t = 0
    while t < max_iter and rel_err > rel_tol:
        rel_err = 0.0
        for q in range(k):
            if is_masked and mask[q] != 0:  # `mask(k) > 0` on an unsigned matrix
                continue
            mu = Wt[q, :] / (Ajt + TINY_NUM)
            a = float(np.dot(Aj, mu * mu))
            b = float(np.dot(Aj, mu)) - sumW[q]
            a += beta[0]
            b += a * Hj[q] - beta[2] - beta[1] * (sumHj - Hj[q])
            tmp = b / (a + TINY_NUM)
            if tmp < 0:
                tmp = 0.0
            if tmp != Hj[q]:
                Ajt += (tmp - Hj[q]) * Wt[q, :]
                etmp = 2 * abs(Hj[q] - tmp) / (tmp + Hj[q] + TINY_NUM)
                if etmp > rel_err:
                    rel_err = etmp
                sumHj += tmp - Hj[q]
                Hj[q] = tmp
        t += 1
    return t


def lee_kl_update(Hj, Wt, Aj, sumW, mask, beta, max_iter, rel_tol):
    """src/base_algorithms.cpp:119-151."""
    k = Wt.shape[0]
    sumHj = float(np.sum(Hj))
    rel_err = rel_tol + 1.0
    is_masked = mask is not None and mask.size > 0
    wh = Wt.T @ Hj
    t = 0
    while t < max_iter and rel_err > rel_tol:
        rel_err = 0.0
        for q in range(k):
            if is_masked and mask[q] != 0:  # `mask(k) > 0` on an unsigned matrix
                continue
            tmp = float(np.dot(Wt[q, :], Aj / (wh + TINY_NUM)))
            tmp /= (sumW[q] + beta[0] * Hj[q] + beta[1] * (sumHj - Hj[q]) + beta[2])
            wh += (tmp - 1) * Hj[q] * Wt[q, :]
            sumHj += (tmp - 1) * Hj[q]
            Hj[q] *= tmp
            tmp = 2 * abs(tmp - 1) / (tmp + 1)
            if tmp > rel_err:
                rel_err = tmp
        t += 1
    return t


# --------------------------------------------------------------------------------------
# half-steps (src/update_with_missing.cpp)
# --------------------------------------------------------------------------------------
def _mask_col(mask, j):
    if mask is None or mask.size == 0:
        return None
    return mask[:, j]


def _gram_edits(WtW, beta):
    """src/update_with_missing.cpp:20-24 / :98-103."""
    if beta[0] != beta[1]:
        WtW[np.diag_indices_from(WtW)] += beta[0] - beta[1]
    if beta[1] != 0:
        WtW += beta[1]
    WtW[np.diag_indices_from(WtW)] += TINY_NUM
    return WtW


def update(H, Wt, A, mask, beta, max_iter, rel_tol, method):
    """Dense half-step, src/update_with_missing.cpp:3-55.  H (k x m) updated in place.

    Solves A ~ Wt^T H for H.  Returns the total number of per-column sweeps.
    """
    m = A.shape[1]
    total = 0
    is_masked = mask is not None and mask.size > 0
    WtW = sumW = None
    if method in (1, 2):
        WtW = _gram_edits(Wt @ Wt.T, beta)
    else:
        sumW = Wt.sum(axis=1)
    for j in range(m):
        if is_masked and np.all(mask[:, j] != 0):
            continue
        mj = _mask_col(mask, j)
        if method == 1:
            mu = WtW @ H[:, j] - Wt @ A[:, j]
            if beta[2] != 0:
                mu += beta[2]
            it = scd_ls_update(H[:, j], WtW, mu, mj, max_iter, rel_tol)
        elif method == 2:
            it = lee_ls_update(H[:, j], WtW, Wt @ A[:, j], beta[2], mj, max_iter, rel_tol)
        elif method == 3:
            it = scd_kl_update(H[:, j], Wt, A[:, j], sumW, mj, beta, max_iter, rel_tol)
        elif method == 4:
            it = lee_kl_update(H[:, j], Wt, A[:, j], sumW, mj, beta, max_iter, rel_tol)
        else:
            it = 0
        total += it
    return total


def update_with_missing(H, Wt, A, mask, beta, max_iter, rel_tol, method):
    """Half-step with NA in A, src/update_with_missing.cpp:58-139."""
    m = A.shape[1]
    total = 0
    is_masked = mask is not None and mask.size > 0
    for j in range(m):
        if is_masked and np.all(mask[:, j] != 0):
            continue
        mj = _mask_col(mask, j)
        col = A[:, j]
        fin = np.isfinite(col)
        any_missing = not bool(fin.all())
        nm = np.nonzero(fin)[0]
        WtW = mu = None
        if method in (1, 2):
            if any_missing:
                Wn = Wt[:, nm]
                WtW = Wn @ Wn.T
                mu = Wn @ col[nm]
            else:
                WtW = Wt @ Wt.T
                mu = Wt @ col
            WtW = _gram_edits(WtW, beta)
        it = 0
        if method == 1:
            mu = WtW @ H[:, j] - mu
            if beta[2] != 0:
                mu += beta[2]
            it = scd_ls_update(H[:, j], WtW, mu, mj, max_iter, rel_tol)
        elif method == 2:
            it = lee_ls_update(H[:, j], WtW, mu, beta[2], mj, max_iter, rel_tol)
        elif method == 3:
            if any_missing:
                Wn = Wt[:, nm]
                it = scd_kl_update(H[:, j], Wn, col[nm], Wn.sum(axis=1), mj, beta, max_iter, rel_tol)
            else:
                it = scd_kl_update(H[:, j], Wt, col, Wt.sum(axis=1), mj, beta, max_iter, rel_tol)
        elif method == 4:
            if any_missing:
                Wn = Wt[:, nm]
                it = lee_kl_update(H[:, j], Wn, col[nm], Wn.sum(axis=1), mj, beta, max_iter, rel_tol)
            else:
                it = lee_kl_update(H[:, j], Wt, col, Wt.sum(axis=1), mj, beta, max_iter, rel_tol)
        total += it
    return total


# --------------------------------------------------------------------------------------
# drivers (src/nnmf.cpp, src/nnlm.cpp)
# --------------------------------------------------------------------------------------
def add_penalty(terr, Wt, H, N_non_missing, alpha, beta):
    """src/nnmf.cpp:224-240 (W there is the k x n transposed factor)."""
    if alpha[0] != alpha[1]:
        terr += 0.5 * (alpha[0] - alpha[1]) * float(np.sum(Wt * Wt)) / N_non_missing
    if beta[0] != beta[1]:
        terr += 0.5 * (beta[0] - beta[1]) * float(np.sum(H * H)) / N_non_missing
    if alpha[1] != 0:
        terr += 0.5 * alpha[1] * float(np.sum(Wt @ Wt.T)) / N_non_missing
    if beta[1] != 0:
        terr += 0.5 * beta[1] * float(np.sum(H @ H.T)) / N_non_missing
    if alpha[2] != 0:
        terr += alpha[2] * float(np.sum(Wt)) / N_non_missing
    if beta[2] != 0:
        terr += beta[2] * float(np.sum(H)) / N_non_missing
    return terr


def _errors(A, Wt, H, fin, any_missing):
    """MSE and the variable part of MKL, src/nnmf.cpp:121-126,135-140."""
    Ahat = Wt.T @ H
    with np.errstate(divide="ignore", invalid="ignore"):
        if any_missing:
            d = (A - Ahat)[fin]
            mse = float(np.mean(d * d))
            kl = float(np.mean((-(A + TINY_NUM) * np.log(Ahat + TINY_NUM) + Ahat)[fin]))
        else:
            d = A - Ahat
            mse = float(np.mean(np.mean(d * d, axis=0)))
            kl = float(np.mean(np.mean(-(A + TINY_NUM) * np.log(Ahat + TINY_NUM) + Ahat, axis=0)))
    return mse, kl


def c_nnmf(A, k, W, H, Wm, Hm, alpha, beta, max_iter, rel_tol, n_threads, verbose,
           show_warning, inner_max_iter, inner_rel_tol, method, trace, rng=None):
    """src/nnmf.cpp:4-220 with the 17-argument signature of src/RcppExports.cpp:29-51.

    ``W`` is n x k or n x 0 (None), ``H`` is k x m or 0 x m (None); ``Wm``/``Hm`` logical
    masks of the same shapes or empty/None.  ``rng`` supplies the default init
    (numpy Generator standing in for R's unif_rand; src/nnmf.cpp:82-98).
    Returns the reference's named list as a dict (+ 'warning': bool).
    """
    A = np.asarray(A, dtype=np.float64)
    n, m = A.shape
    k = int(k)
    max_iter = int(max_iter) & 0xFFFFFFFF
    trace = int(trace) & 0xFFFFFFFF
    if trace < 1:
        trace = 1
    err_len = int(math.ceil(float(max_iter) / float(trace))) + 1
    mse_err = np.zeros(err_len)
    mkl_err = np.zeros(err_len)
    terr = np.zeros(err_len)
    ave_epoch = np.zeros(err_len)

    rel_err = rel_tol + 1.0
    terr_last = 1e99
    fin = np.isfinite(A)
    any_missing = not bool(fin.all())
    N_non_missing = n * m
    with np.errstate(divide="ignore", invalid="ignore"):
        if any_missing:
            N_non_missing = int(fin.sum())
            a = A[fin]
            mkl_err[:] = float(np.mean((a + TINY_NUM) * np.log(a + TINY_NUM) - a))
        else:
            mkl_err[:] = float(np.mean(np.mean((A + TINY_NUM) * np.log(A + TINY_NUM) - A, axis=0)))

    Wmt = None if Wm is None or np.size(Wm) == 0 else np.ascontiguousarray(np.asarray(Wm).T).astype(np.int64)
    Hmm = None if Hm is None or np.size(Hm) == 0 else np.asarray(Hm).astype(np.int64)

    if W is None or np.size(W) == 0:
        rng = rng or np.random.default_rng(0)
        Wt = rng.random((n, k)).T.copy() * 0.01  # column-major draw order of a k x n matrix
        if Wmt is not None:
            Wt[Wmt != 0] = 0.0
    else:
        Wt = np.array(np.asarray(W, dtype=np.float64).T, order="C", copy=True)
    if H is None or np.size(H) == 0:
        rng = rng or np.random.default_rng(0)
        Hc = rng.random((m, k)).T.copy() * 0.01
        if Hmm is not None:
            Hc[Hmm != 0] = 0.0
    else:
        Hc = np.array(H, dtype=np.float64, copy=True)

    upd = update_with_missing if any_missing else update
    At = np.ascontiguousarray(A.T)
    total_raw_iter = 0
    i = 0
    i_e = 0

    def error_block(i_e, total_raw_iter, terr_last):
        mse, kl = _errors(A, Wt, Hc, fin, any_missing)
        mse_err[i_e] = mse
        mkl_err[i_e] += kl
        ave_epoch[i_e] = float(total_raw_iter) / (n + m)
        t = 0.5 * mse_err[i_e] if method < 3 else mkl_err[i_e]
        t = add_penalty(t, Wt, Hc, N_non_missing, alpha, beta)
        terr[i_e] = t
        rel = 2 * (terr_last - t) / (terr_last + t + TINY_NUM)
        return rel, t

    while i < max_iter and abs(rel_err) > rel_tol:
        total_raw_iter += upd(Wt, Hc, At, Wmt, alpha, inner_max_iter, inner_rel_tol, method)
        total_raw_iter += upd(Hc, Wt, A, Hmm, beta, inner_max_iter, inner_rel_tol, method)
        if i % trace == 0:
            rel_err, terr_last = error_block(i_e, total_raw_iter, terr_last)
            total_raw_iter = 0
            i_e += 1
        i += 1

    if ((i - 1) & 0xFFFFFFFF) % trace != 0:  # unsigned arithmetic, src/nnmf.cpp:164
        rel_err, terr_last = error_block(i_e, total_raw_iter, terr_last)
        i_e += 1

    warn = bool(show_warning and rel_err > rel_tol)  # src/nnmf.cpp:208 (no abs)
    return dict(W=Wt.T.copy(), H=Hc, mse_error=mse_err[:i_e].copy(), mkl_error=mkl_err[:i_e].copy(),
                target_error=terr[:i_e].copy(), average_epoch=ave_epoch[:i_e].copy(),
                n_iteration=i, warning=warn)


def c_nnlm(x, y, alpha, mask, beta0, max_iter, rel_tol, n_threads, method, rng=None):
    """src/nnlm.cpp:4-53.  x n x p, y n x q, mask p x q or empty, beta0 p x q or empty."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if y.ndim == 1:
        y = y[:, None]
    p, q = x.shape[1], y.shape[1]
    any_missing = not bool(np.isfinite(y).all())
    if beta0 is None or np.size(beta0) == 0:
        rng = rng or np.random.default_rng(0)
        beta = rng.random((q, p)).T.copy()
    else:
        beta = np.array(beta0, dtype=np.float64, copy=True).reshape(p, q)
    mk = None if mask is None or np.size(mask) == 0 else np.asarray(mask).astype(np.int64).reshape(p, q)
    xt = np.ascontiguousarray(x.T)
    upd = update_with_missing if any_missing else update
    nstep = upd(beta, xt, y, mk, alpha, max_iter, rel_tol, method)
    return dict(coefficient=beta, n_iteration=nstep)
