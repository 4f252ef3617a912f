"""ctypes binding of libnnlm_mi355x.so (include/nnlm_mi355x.h).

This is the Python stand-in for the R-side ``.Call`` stub (pkg/src/r_glue.c): it passes
plain pointers and sizes across the C ABI, nothing else.  There is deliberately NO fallback: if
the shared library is missing or no gfx950 device is present, every entry raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnnlm_mi355x.so")

NNLM_OK = 0
PREC_F32 = 0
PREC_F64 = 1
COMM_ID_BYTES = 128
FORM_COLS = 0    # column-sharded half-steps, one all-gather each (default)
FORM_REDUCE = 1  # dense square loss: contraction-sharded [G | C] + all-reduce, then sweep + all-gather
FORMS = {"cols": FORM_COLS, "reduce": FORM_REDUCE}

# every symbol include/nnlm_mi355x.h declares (tests check the .so exports all of them)
EXPORTS = [
    "nnlm_trace_capacity", "nnlm_c_nnmf", "nnlm_c_nnlm", "nnlm_create", "nnlm_destroy", "nnlm_last_error",
    "nnlm_abi_version", "nnlm_set_matrix", "nnlm_matrix_info", "nnlm_set_factors", "nnlm_get_factors",
    "nnlm_half_step", "nnlm_iterate", "nnlm_run", "nnlm_take_sweeps", "nnlm_errors", "nnlm_sync", "nnlm_profile_enable",
    "nnlm_profile_get", "nnlm_profile_reset", "nnlm_comm_unique_id", "nnlm_comm_init", "nnlm_comm_info",
    "nnlm_shard_range", "nnlm_shard_cols", "nnlm_debug_partial", "nnlm_debug_phase", "nnlm_debug_exchange",
    "nnlm_comm_set_form", "nnlm_debug_set_cus", "nnlm_get_info", "nnlm_debug_alloc_limit", "nnlm_release_caches",
]


class NnlmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libnnlm_mi355x error {code}: {msg}")
        self.code = code


class Callbacks(C.Structure):
    _fields_ = [
        ("ctx", C.c_void_p),
        ("check_interrupt", C.CFUNCTYPE(C.c_int, C.c_void_p)),
        ("progress", C.CFUNCTYPE(None, C.c_void_p, C.c_uint, C.c_uint)),
        ("print", C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)),
        ("warning", C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)),
        ("unif_rand", C.CFUNCTYPE(C.c_double, C.c_void_p)),
    ]


_lib = None


def load():
    """Load the shared library (raises if it has not been built: there is no CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NnlmError(-1, f"{LIB_PATH} not found: build it with `python -m nnlm_amd.build` (hipcc, gfx950)")
    lib = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    lib.nnlm_abi_version.restype = C.c_int
    lib.nnlm_last_error.restype = C.c_char_p
    lib.nnlm_last_error.argtypes = [vp]
    lib.nnlm_trace_capacity.restype = C.c_uint
    lib.nnlm_trace_capacity.argtypes = [C.c_uint, C.c_uint]
    lib.nnlm_c_nnmf.restype = C.c_int
    lib.nnlm_c_nnmf.argtypes = [dp, C.c_int, C.c_int, C.c_uint, dp, dp, ip, ip, dp, dp, C.c_uint, C.c_double, C.c_int,
                                C.c_int, C.c_int, C.c_uint, C.c_double, C.c_int, C.c_uint, dp, dp, dp, dp, dp, dp, ip,
                                C.POINTER(C.c_uint), ip, C.POINTER(Callbacks)]
    lib.nnlm_c_nnlm.restype = C.c_int
    lib.nnlm_c_nnlm.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int, dp, ip, dp, C.c_uint, C.c_double, C.c_int, C.c_int,
                                dp, ip, C.POINTER(Callbacks)]
    lib.nnlm_create.restype = C.c_int
    lib.nnlm_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    lib.nnlm_destroy.restype = None
    lib.nnlm_destroy.argtypes = [vp]
    lib.nnlm_set_matrix.restype = C.c_int
    lib.nnlm_set_matrix.argtypes = [vp, dp, C.c_int, C.c_int]
    lib.nnlm_matrix_info.restype = C.c_int
    lib.nnlm_matrix_info.argtypes = [vp, dp, ip, dp]
    lib.nnlm_set_factors.restype = C.c_int
    lib.nnlm_set_factors.argtypes = [vp, C.c_uint, dp, dp, ip, ip]
    lib.nnlm_get_factors.restype = C.c_int
    lib.nnlm_get_factors.argtypes = [vp, dp, dp]
    lib.nnlm_half_step.restype = C.c_int
    lib.nnlm_half_step.argtypes = [vp, C.c_int, dp, C.c_uint, C.c_double, C.c_int]
    lib.nnlm_iterate.restype = C.c_int
    lib.nnlm_iterate.argtypes = [vp, C.c_uint, dp, dp, C.c_uint, C.c_double, C.c_int]
    lib.nnlm_run.restype = C.c_int
    lib.nnlm_run.argtypes = [vp, dp, dp, C.c_uint, C.c_double, C.c_int, C.c_int, C.c_uint, C.c_double, C.c_int, C.c_uint,
                             dp, dp, dp, dp, ip, C.POINTER(C.c_uint), ip, C.POINTER(Callbacks)]
    lib.nnlm_take_sweeps.restype = C.c_int
    lib.nnlm_take_sweeps.argtypes = [vp, C.POINTER(C.c_longlong), C.c_int]
    lib.nnlm_errors.restype = C.c_int
    lib.nnlm_errors.argtypes = [vp, dp, dp, dp]
    lib.nnlm_sync.restype = C.c_int
    lib.nnlm_sync.argtypes = [vp]
    lib.nnlm_profile_enable.restype = C.c_int
    lib.nnlm_profile_enable.argtypes = [vp, C.c_int]
    lib.nnlm_profile_reset.restype = C.c_int
    lib.nnlm_profile_reset.argtypes = [vp]
    lib.nnlm_profile_get.restype = C.c_int
    lib.nnlm_profile_get.argtypes = [vp, C.c_char_p, dp, C.POINTER(C.c_longlong)]
    lib.nnlm_comm_unique_id.restype = C.c_int
    lib.nnlm_comm_unique_id.argtypes = [C.c_char_p]
    lib.nnlm_comm_init.restype = C.c_int
    lib.nnlm_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    lib.nnlm_comm_info.restype = C.c_int
    lib.nnlm_comm_info.argtypes = [vp, ip, ip]
    lib.nnlm_shard_range.restype = C.c_int
    lib.nnlm_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, ip]
    lib.nnlm_shard_cols.restype = C.c_int
    lib.nnlm_shard_cols.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, ip]
    lib.nnlm_debug_partial.restype = C.c_int
    lib.nnlm_debug_partial.argtypes = [vp, C.c_int, dp, dp]
    lib.nnlm_debug_phase.restype = C.c_int
    lib.nnlm_debug_phase.argtypes = [vp, C.c_int, C.c_int, dp, C.c_uint, C.c_double, C.c_int]
    lib.nnlm_debug_exchange.restype = C.c_int
    lib.nnlm_debug_exchange.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int]
    lib.nnlm_comm_set_form.restype = C.c_int
    lib.nnlm_comm_set_form.argtypes = [vp, C.c_int]
    lib.nnlm_debug_set_cus.restype = C.c_int
    lib.nnlm_debug_set_cus.argtypes = [C.c_int]
    lib.nnlm_debug_alloc_limit.restype = C.c_int
    lib.nnlm_debug_alloc_limit.argtypes = [C.c_size_t]
    lib.nnlm_release_caches.restype = C.c_int
    lib.nnlm_release_caches.argtypes = []
    lib.nnlm_get_info.restype = C.c_int
    lib.nnlm_get_info.argtypes = [vp, C.c_char_p, dp]
    _lib = lib
    return lib


def _check(rc, handle=None):
    if rc != NNLM_OK:
        msg = load().nnlm_last_error(handle)
        raise NnlmError(rc, msg.decode() if msg else "unknown")


def _f64(a, shape=None):
    """Column-major fp64 array (what R hands to .Call).  The library never writes its inputs, so an array that already is fp64 and
    Fortran-contiguous is passed as it is (a 1.6 GB transposing copy of a C-ordered 20000 x 10000 matrix takes longer than its upload)."""
    arr = np.asarray(a, dtype=np.float64)
    if shape is not None:
        arr = arr.reshape(shape, order="F") if arr.flags.f_contiguous else np.asfortranarray(arr).reshape(shape, order="F")
    if not arr.flags.f_contiguous:
        arr = np.asfortranarray(arr)
    return arr


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def _lgl(mask, shape):
    """R logical matrix -> int32 column-major, or None when empty (src/nnmf.cpp:75-80)."""
    if mask is None or np.size(mask) == 0:
        return None
    return np.array(np.asarray(mask).reshape(shape) != 0, dtype=np.int32, order="F")


def _vec3(v):
    return np.array(v, dtype=np.float64).reshape(3)


def make_callbacks(unif_rand=None, print_fn=None, warning=None, progress=None, check_interrupt=None):
    """Build an nnlm_callbacks struct from Python callables (kept alive by the returned object)."""
    cb = Callbacks()
    keep = []

    def wrap(ftype, fn, adapt):
        if fn is None:
            return ftype()
        f = ftype(adapt(fn))
        keep.append(f)
        return f

    fields = dict(Callbacks._fields_)
    cb.check_interrupt = wrap(fields["check_interrupt"], check_interrupt, lambda fn: (lambda ctx: int(bool(fn()))))
    cb.progress = wrap(fields["progress"], progress, lambda fn: (lambda ctx, d, t: fn(d, t)))
    cb.print = wrap(fields["print"], print_fn, lambda fn: (lambda ctx, s: fn(s.decode())))
    cb.warning = wrap(fields["warning"], warning, lambda fn: (lambda ctx, s: fn(s.decode())))
    cb.unif_rand = wrap(fields["unif_rand"], unif_rand, lambda fn: (lambda ctx: float(fn())))
    cb._keep = keep
    return cb


# ----------------------------------------------------------------------------------------------
# one-shot entries: same argument lists as the reference's c_nnmf / c_nnlm
# ----------------------------------------------------------------------------------------------
def c_nnmf(A, k, W, H, Wm, Hm, alpha, beta, max_iter, rel_tol, n_threads, verbose, show_warning,
           inner_max_iter, inner_rel_tol, method, trace, callbacks=None):
    """.Call('_NNLM_c_nnmf', ...) equivalent (reference src/RcppExports.cpp:29-54) -> named list as dict."""
    lib = load()
    A = _f64(A)
    n, m = A.shape
    k = int(k)
    Wi = _f64(W, (n, k)) if W is not None and np.size(W) > 0 else None
    Hi = _f64(H, (k, m)) if H is not None and np.size(H) > 0 else None
    Wm_, Hm_ = _lgl(Wm, (n, k)), _lgl(Hm, (k, m))
    al, be = _vec3(alpha), _vec3(beta)
    cap = lib.nnlm_trace_capacity(int(max_iter), int(trace) if int(trace) > 0 else 1)
    Wo = np.zeros((n, k), order="F")
    Ho = np.zeros((k, m), order="F")
    mse, mkl, terr, ep = (np.zeros(cap) for _ in range(4))
    n_trace, n_it, warned = C.c_int(0), C.c_uint(0), C.c_int(0)
    rc = lib.nnlm_c_nnmf(_dp(A), n, m, k, _dp(Wi), _dp(Hi), _ip(Wm_), _ip(Hm_), _dp(al), _dp(be), int(max_iter),
                         float(rel_tol), int(n_threads), int(verbose), int(bool(show_warning)), int(inner_max_iter),
                         float(inner_rel_tol), int(method), int(trace) & 0xFFFFFFFF, _dp(Wo), _dp(Ho), _dp(mse), _dp(mkl),
                         _dp(terr), _dp(ep), C.byref(n_trace), C.byref(n_it), C.byref(warned),
                         C.byref(callbacks) if callbacks is not None else None)
    _check(rc)
    e = n_trace.value
    return dict(W=np.ascontiguousarray(Wo), H=np.ascontiguousarray(Ho), mse_error=mse[:e].copy(), mkl_error=mkl[:e].copy(),
                target_error=terr[:e].copy(), average_epoch=ep[:e].copy(), n_iteration=int(n_it.value),
                warning=bool(warned.value))


def c_nnlm(x, y, alpha, mask, beta0, max_iter, rel_tol, n_threads, method, callbacks=None):
    """.Call('_NNLM_c_nnlm', ...) equivalent (reference src/RcppExports.cpp:10-27)."""
    lib = load()
    x = _f64(x)
    n, p = x.shape
    y = _f64(np.asarray(y, dtype=np.float64).reshape(n, -1))
    q = y.shape[1]
    b0 = _f64(beta0, (p, q)) if beta0 is not None and np.size(beta0) > 0 else None
    mk = _lgl(mask, (p, q))
    al = _vec3(alpha)
    coef = np.zeros((p, q), order="F")
    nit = C.c_int(0)
    rc = lib.nnlm_c_nnlm(_dp(x), _dp(y), n, p, q, _dp(al), _ip(mk), _dp(b0), int(max_iter), float(rel_tol),
                         int(n_threads), int(method), _dp(coef), C.byref(nit),
                         C.byref(callbacks) if callbacks is not None else None)
    _check(rc)
    return dict(coefficient=np.ascontiguousarray(coef), n_iteration=int(nit.value))


# ----------------------------------------------------------------------------------------------
# resident API
# ----------------------------------------------------------------------------------------------
class Handle:
    """Device-resident problem (A stays in HBM across calls)."""

    def __init__(self, device=0, precision=PREC_F32):
        self._lib = load()
        self._h = C.c_void_p()
        _check(self._lib.nnlm_create(C.byref(self._h), int(device), int(precision)))
        self.n = self.m = self.k = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nnlm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        _check(rc, self._h)

    def set_matrix(self, A):
        A = _f64(A)
        self.n, self.m = A.shape
        self._ck(self._lib.nnlm_set_matrix(self._h, _dp(A), self.n, self.m))

    def matrix_info(self):
        nn, am, kc = C.c_double(0), C.c_int(0), C.c_double(0)
        self._ck(self._lib.nnlm_matrix_info(self._h, C.byref(nn), C.byref(am), C.byref(kc)))
        return dict(n_non_missing=nn.value, any_missing=bool(am.value), kl_const=kc.value)

    def set_factors(self, k, W=None, H=None, Wm=None, Hm=None):
        self.k = int(k)
        Wi = _f64(W, (self.n, self.k)) if W is not None else None
        Hi = _f64(H, (self.k, self.m)) if H is not None else None
        self._ck(self._lib.nnlm_set_factors(self._h, self.k, _dp(Wi), _dp(Hi), _ip(_lgl(Wm, (self.n, self.k))),
                                            _ip(_lgl(Hm, (self.k, self.m)))))

    def get_factors(self):
        W = np.zeros((self.n, self.k), order="F")
        H = np.zeros((self.k, self.m), order="F")
        self._ck(self._lib.nnlm_get_factors(self._h, _dp(W), _dp(H)))
        return np.ascontiguousarray(W), np.ascontiguousarray(H)

    def half_step(self, which, reg, inner_max_iter, inner_rel_tol, method):
        r = _vec3(reg)
        self._ck(self._lib.nnlm_half_step(self._h, int(which), _dp(r), int(inner_max_iter), float(inner_rel_tol), int(method)))

    def iterate(self, n_iter, alpha, beta, inner_max_iter, inner_rel_tol, method):
        a, b = _vec3(alpha), _vec3(beta)
        self._ck(self._lib.nnlm_iterate(self._h, int(n_iter), _dp(a), _dp(b), int(inner_max_iter), float(inner_rel_tol), int(method)))

    def run(self, alpha, beta, max_iter, rel_tol, verbose, show_warning, inner_max_iter, inner_rel_tol, method, trace,
            callbacks=None):
        """The resident c_nnmf loop (reference src/nnmf.cpp:100-209); returns the traces as a dict."""
        a, b = _vec3(alpha), _vec3(beta)
        cap = self._lib.nnlm_trace_capacity(int(max_iter), int(trace) if int(trace) > 0 else 1)
        mse, mkl, terr, ep = (np.zeros(cap) for _ in range(4))
        n_trace, n_it, warned = C.c_int(0), C.c_uint(0), C.c_int(0)
        self._ck(self._lib.nnlm_run(self._h, _dp(a), _dp(b), int(max_iter), float(rel_tol), int(verbose), int(bool(show_warning)),
                                    int(inner_max_iter), float(inner_rel_tol), int(method), int(trace) & 0xFFFFFFFF, _dp(mse),
                                    _dp(mkl), _dp(terr), _dp(ep), C.byref(n_trace), C.byref(n_it), C.byref(warned),
                                    C.byref(callbacks) if callbacks is not None else None))
        e = n_trace.value
        return dict(mse_error=mse[:e].copy(), mkl_error=mkl[:e].copy(), target_error=terr[:e].copy(),
                    average_epoch=ep[:e].copy(), n_iteration=int(n_it.value), warning=bool(warned.value))

    def take_sweeps(self, reset=True):
        v = C.c_longlong(0)
        self._ck(self._lib.nnlm_take_sweeps(self._h, C.byref(v), int(reset)))
        return int(v.value)

    def errors(self):
        mse, kl = C.c_double(0), C.c_double(0)
        pen = np.zeros(6)
        self._ck(self._lib.nnlm_errors(self._h, C.byref(mse), C.byref(kl), _dp(pen)))
        return mse.value, kl.value, pen

    def sync(self):
        self._ck(self._lib.nnlm_sync(self._h))

    def profile_enable(self, on=True):
        self._ck(self._lib.nnlm_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._ck(self._lib.nnlm_profile_reset(self._h))

    def profile_get(self, name):
        ms, cnt = C.c_double(0), C.c_longlong(0)
        self._ck(self._lib.nnlm_profile_get(self._h, name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, int(cnt.value)

    def comm_init(self, unique_id, rank: int, nranks: int, form="cols"):
        """unique_id=None -> virtual rank (no communicator; partial sums stay un-reduced).  form: "cols" (column-sharded half-steps,
        one all-gather each) or "reduce" (dense square loss: contraction-sharded + all-reduce, then sweep + all-gather)."""
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES) if unique_id is not None else None
        self._ck(self._lib.nnlm_comm_init(self._h, buf, int(rank), int(nranks)))
        self.comm_set_form(form)

    def comm_set_form(self, form):
        self._ck(self._lib.nnlm_comm_set_form(self._h, FORMS[form] if isinstance(form, str) else int(form)))

    def get_info(self, key):
        """cus, sweep_form_w / sweep_form_h (0 plain, 1 persistent -- strict fp64 --, 2 fp32 chain, -1 none yet), sweep_groups_w / sweep_groups_h."""
        v = C.c_double(0)
        self._ck(self._lib.nnlm_get_info(self._h, key.encode(), C.byref(v)))
        return v.value

    def debug_phase(self, which, phase, reg, inner_max_iter, inner_rel_tol, method):
        r = _vec3(reg)
        self._ck(self._lib.nnlm_debug_phase(self._h, int(which), int(phase), _dp(r), int(inner_max_iter), float(inner_rel_tol), int(method)))

    def debug_partial(self, which):
        cols = self.m if which == 1 else self.n
        G = np.zeros((self.k, self.k), order="F")
        Cm = np.zeros((self.k, cols), order="F")
        self._ck(self._lib.nnlm_debug_partial(self._h, int(which), _dp(G), _dp(Cm)))
        return np.ascontiguousarray(G), np.ascontiguousarray(Cm)

    def comm_info(self):
        r, n = C.c_int(0), C.c_int(0)
        self._ck(self._lib.nnlm_comm_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value


def shard_range(n, m, precision, which, rank, nranks):
    """Contraction range [begin, end) of `rank` (pure host function of the C ABI, works without a GPU)."""
    b, e = C.c_int(0), C.c_int(0)
    _check(load().nnlm_shard_range(int(n), int(m), int(precision), int(which), int(rank), int(nranks), C.byref(b), C.byref(e)))
    return b.value, e.value


def shard_cols(ncols, rank, nranks):
    """(cpr, col0, col1): the columns `rank` solves and the packed slab width (pure host function of the C ABI)."""
    cpr, c0, c1 = C.c_int(0), C.c_int(0), C.c_int(0)
    _check(load().nnlm_shard_cols(int(ncols), int(rank), int(nranks), C.byref(cpr), C.byref(c0), C.byref(c1)))
    return cpr.value, c0.value, c1.value


def debug_exchange(handles, which, stage):
    """Host stand-in for ncclAllReduce (stage 1) / ncclAllGather (stage 2) between virtual ranks (test hook)."""
    arr = (C.c_void_p * len(handles))(*[h._h for h in handles])
    _check(load().nnlm_debug_exchange(arr, len(handles), int(which), int(stage)))


def debug_set_cus(cus: int):
    """Test hook: handles created from now on plan their launches for `cus` compute units (0 = the device's own count)."""
    _check(load().nnlm_debug_set_cus(int(cus)))


def debug_alloc_limit(nbytes: int):
    """Test hook: matrix-sized KL workspaces beyond `nbytes` "do not fit" (0 = no limit) -> the streaming path over column chunks."""
    _check(load().nnlm_debug_alloc_limit(int(nbytes)))


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load().nnlm_comm_unique_id(buf))
    return buf.raw


def release_caches():
    """Release the process-wide caches of the library (resources of the last destroyed handle, pinned bounce buffers)."""
    _check(load().nnlm_release_caches())
