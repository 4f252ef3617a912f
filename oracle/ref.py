"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/libnnlm_ref.so (oracle/nnlm_ref.c).

Mirrors the reference's 17-argument ``c_nnmf`` / 9-argument ``c_nnlm`` .Call entry points
(src/RcppExports.cpp:10-51) with numpy arrays.  Never imported by the product package.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libnnlm_ref.so")
    src = os.path.join(_HERE, "nnlm_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnnlm_ref.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        _LIB.ref_c_nnmf.restype = C.c_int
        _LIB.ref_c_nnmf.argtypes = [dp, C.c_int, C.c_int, C.c_uint, dp, C.c_int, dp, C.c_int, ip, ip, dp, dp,
                                    C.c_uint, C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_double,
                                    C.c_int, C.c_uint, dp, dp, dp, dp, ip, C.POINTER(C.c_uint), ip, C.c_void_p]
        _LIB.ref_c_nnlm.restype = C.c_int
        _LIB.ref_c_nnlm.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int, dp, ip, dp, C.c_int, C.c_uint, C.c_double,
                                    C.c_int, C.c_int, C.c_void_p]
        _LIB.ref_update.restype = C.c_int
        _LIB.ref_update.argtypes = [dp, dp, dp, ip, dp, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_double, C.c_int, C.c_int]
        _LIB.ref_update_with_missing.restype = C.c_int
        _LIB.ref_update_with_missing.argtypes = _LIB.ref_update.argtypes
    return _LIB


def _f(a):  # column-major fp64 copy
    return np.array(a, dtype=np.float64, order="F", copy=True)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def _mask(mk, shape):
    if mk is None or np.size(mk) == 0:
        return None
    return np.array(np.asarray(mk).reshape(shape) != 0, dtype=np.int32, order="F")


def update(H, Wt, A, mask, beta, max_iter, rel_tol, method, n_threads=0, missing=None):
    """One half-step (src/update_with_missing.cpp:3-55 or :58-139). Returns (H_new, sweeps)."""
    H = _f(H); Wt = _f(Wt); A = _f(A)
    k, m = H.shape
    n = Wt.shape[1]
    assert A.shape == (n, m) and Wt.shape[0] == k
    mk = _mask(mask, (k, m))
    b = np.array(beta, dtype=np.float64)
    if missing is None:
        missing = not bool(np.isfinite(A).all())
    fn = lib().ref_update_with_missing if missing else lib().ref_update
    it = fn(_dp(H), _dp(Wt), _dp(A), _ip(mk), _dp(b), k, n, m, int(max_iter), float(rel_tol), int(n_threads), int(method))
    return H, it


def c_nnmf(A, k, W, H, Wm, Hm, alpha, beta, max_iter, rel_tol, n_threads, verbose, show_warning,
           inner_max_iter, inner_rel_tol, method, trace):
    """Same argument list and named result as the reference's c_nnmf (src/nnmf.cpp:4-7,211-219)."""
    A = _f(A)
    n, m = A.shape
    k = int(k)
    W_given = W is not None and np.size(W) > 0
    H_given = H is not None and np.size(H) > 0
    Wb = _f(W).reshape(n, k, order="F") if W_given else np.zeros((n, k), order="F")
    Hb = _f(H).reshape(k, m, order="F") if H_given else np.zeros((k, m), order="F")
    Wm_ = _mask(Wm, (n, k)); Hm_ = _mask(Hm, (k, m))
    al = np.array(alpha, dtype=np.float64); be = np.array(beta, dtype=np.float64)
    tr = max(int(trace), 1)
    err_len = int(math.ceil(float(max_iter) / float(tr))) + 1
    mse = np.zeros(err_len); mkl = np.zeros(err_len); terr = np.zeros(err_len); ep = np.zeros(err_len)
    n_err = C.c_int(0); n_it = C.c_uint(0); warn = C.c_int(0)
    rc = lib().ref_c_nnmf(_dp(A), n, m, k, _dp(Wb), int(W_given), _dp(Hb), int(H_given), _ip(Wm_), _ip(Hm_),
                          _dp(al), _dp(be), int(max_iter), float(rel_tol), int(n_threads), int(verbose),
                          int(bool(show_warning)), int(inner_max_iter), float(inner_rel_tol), int(method), int(trace),
                          _dp(mse), _dp(mkl), _dp(terr), _dp(ep), C.byref(n_err), C.byref(n_it), C.byref(warn), None)
    assert rc == 0
    e = n_err.value
    return dict(W=np.ascontiguousarray(Wb), H=np.ascontiguousarray(Hb), mse_error=mse[:e].copy(),
                mkl_error=mkl[:e].copy(), target_error=terr[:e].copy(), average_epoch=ep[:e].copy(),
                n_iteration=int(n_it.value), warning=bool(warn.value))


def c_nnlm(x, y, alpha, mask, beta0, max_iter, rel_tol, n_threads, method):
    """Same argument list and named result as the reference's c_nnlm (src/nnlm.cpp:4-53)."""
    x = _f(x)
    y = _f(np.asarray(y, dtype=np.float64).reshape(x.shape[0], -1))
    n, p = x.shape
    q = y.shape[1]
    given = beta0 is not None and np.size(beta0) > 0
    beta = _f(np.asarray(beta0, dtype=np.float64).reshape(p, q)) if given else np.zeros((p, q), order="F")
    mk = _mask(mask, (p, q))
    al = np.array(alpha, dtype=np.float64)
    it = lib().ref_c_nnlm(_dp(x), _dp(y), n, p, q, _dp(al), _ip(mk), _dp(beta), int(given), int(max_iter),
                          float(rel_tol), int(n_threads), int(method), None)
    return dict(coefficient=np.ascontiguousarray(beta), n_iteration=int(it))
