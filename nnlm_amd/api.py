"""Host-side mirror of the reference's R interface for the nnmf()/nnlm() path.

Same names, argument meaning, defaults, error and warning behaviour as the R functions, so that the
parity tests read like the reference's own tests:

    nnmf()          <- R/nnmf.R:135-225
    nnlm()          <- R/nnlm.R:70-145
    predict_nnmf()  <- R/nnmf_methods.R:22-48   (S3 predict.nnmf)
    mse_mkl()       <- R/misc.R:9-16
    reformat_input(), get_method_code(), check_matrix() <- R/misc.R:28-129

Compute goes through the C ABI of libnnlm_mi355x.so (nnlm_amd._lib) and nowhere else: there is no
CPU fallback.  The argument normalisation (``prepare_*``) and result decoration (``finish_*``) are
pure host logic and are split out so they can be unit-tested without a GPU.
"""
from __future__ import annotations

import time
import warnings

import numpy as np

from . import _lib


class NnlmStop(ValueError):
    """R's stop()."""


# ------------------------------------------------------------------------------------------------
# R/misc.R
# ------------------------------------------------------------------------------------------------
def mse_mkl(obs, pred, na_rm=True, show_warning=True):
    """R/misc.R:9-16 -> dict(MSE=, MKL=)."""
    obs = np.asarray(obs, dtype=np.float64)
    pred = np.asarray(pred, dtype=np.float64)
    mean = np.nanmean if na_rm else np.mean
    with np.errstate(divide="ignore", invalid="ignore"):
        if (not show_warning) and (np.any(obs < 0) or np.any(pred < 0)):
            mkl = float("nan")
        else:
            mkl = float(mean((obs + 1e-16) * np.log((obs + 1e-16) / (pred + 1e-16)) - obs + pred))
        mse = float(mean((obs - pred) ** 2))
    return {"MSE": mse, "MKL": mkl}


def _match_arg(value, choices, name):
    """R's match.arg for a scalar string (a tuple/list default selects its first entry)."""
    if isinstance(value, (tuple, list)):
        value = value[0]
    hits = [c for c in choices if c.startswith(str(value))]
    if len(hits) != 1:
        raise NnlmStop(f"'{name}' should be one of {', '.join(repr(c) for c in choices)}")
    return hits[0]


def get_method_code(method="scd", loss="mse"):
    """R/misc.R:28-35: 1 scd+mse, 2 lee+mse, 3 scd+mkl, 4 lee+mkl."""
    method = _match_arg(method, ("scd", "lee"), "method")
    loss = _match_arg(loss, ("mse", "mkl"), "loss")
    code = 1
    if loss == "mkl":
        code += 2
    if method == "lee":
        code += 1
    return code


def _is_empty(x):
    return x is None or np.size(x) == 0


def _as_matrix(x):
    """R's as.matrix(): a vector becomes a one-column matrix."""
    a = np.asarray(x)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    elif a.ndim == 1:
        a = a.reshape(-1, 1)
    return a


def check_matrix(A, dm=None, mode="numeric", check_na=False, input_name="", check_negative=False):
    """R/misc.R:38-45."""
    if A is None:
        return
    A = np.asarray(A)
    if dm is not None:
        shp = A.shape if A.ndim == 2 else (A.shape[0] if A.ndim else 1, 1)
        bad = any(e is not None and int(e) != int(s) for s, e in zip(shp, dm))
        if bad:
            raise NnlmStop("Dimension of matrix %s is expected to be (%d, %d), but got (%d, %d)" % (
                input_name, shp[0], shp[1], -1 if dm[0] is None else dm[0], -1 if dm[1] is None else dm[1]))
    is_logical = A.dtype == np.bool_
    if mode == "logical" and not is_logical:
        raise NnlmStop("Matrix %s must be %s." % (input_name, mode))
    if mode == "numeric" and (is_logical or not np.issubdtype(A.dtype, np.number)):
        raise NnlmStop("Matrix %s must be %s." % (input_name, mode))
    if check_negative and mode == "numeric":
        v = A[~np.isnan(A.astype(np.float64))]
        if np.any(v < 0):
            raise NnlmStop("Matrix %s must be non-negative." % input_name)
    if check_na and mode == "numeric" and np.any(np.isnan(A.astype(np.float64))):
        raise NnlmStop("Matrix %s contains missing values." % input_name)


def reformat_input(init, mask, n, m, k, rng=None):
    """R/misc.R:48-129: stack [W W0 W1] / [H; H1; H0] and their masks.

    ``rng`` (numpy Generator) stands in for R's runif() used for blocks that are not supplied while
    a sibling block is (R/misc.R:107-112).
    """
    mask = {} if mask is None else dict(mask)
    init = {} if init is None else dict(init)
    if not isinstance(mask, dict) or not isinstance(init, dict):
        raise NnlmStop("init and mask must be lists (dicts)")
    rng = rng or np.random.default_rng()
    known_W = init.get("W0") is not None
    known_H = init.get("H0") is not None
    kW0 = kH0 = 0
    if known_W:
        init["W0"] = _as_matrix(init["W0"])
        kW0 = init["W0"].shape[1]
        mask["W0"] = np.ones((n, kW0), dtype=bool)
    else:
        mask["W0"] = None
        mask["H1"] = None
        init["H1"] = None
    if known_H:
        init["H0"] = _as_matrix(init["H0"])
        kH0 = init["H0"].shape[0]
        mask["H0"] = np.ones((kH0, m), dtype=bool)
    else:
        mask["H0"] = None
        mask["W1"] = None
        init["W1"] = None
    K = k + kW0 + kH0

    def dims(ew, eh):
        return {"W": (n, k * ew), "W0": (n, kW0 * ew), "W1": (n, kH0 * ew),
                "H": (k * eh, m), "H1": (kW0 * eh, m), "H0": (kH0 * eh, m)}

    ew = int(not all(_is_empty(mask.get(b)) for b in ("W", "W0", "W1")))
    eh = int(not all(_is_empty(mask.get(b)) for b in ("H", "H0", "H1")))
    dm = dims(ew, eh)
    for b in ("W", "W0", "W1", "H", "H0", "H1"):
        if not _is_empty(mask.get(b)):
            check_matrix(mask[b], dm[b], "logical", True, "mask$" + b)
            mask[b] = np.asarray(mask[b], dtype=bool).reshape(dm[b])
        else:
            mask[b] = np.zeros(dm[b], dtype=bool)
    ew = int(not all(_is_empty(init.get(b)) for b in ("W", "W0", "W1")))
    eh = int(not all(_is_empty(init.get(b)) for b in ("H", "H0", "H1")))
    di = dims(ew, eh)
    for b in ("W", "W0", "W1", "H", "H0", "H1"):
        if not _is_empty(init.get(b)):
            check_matrix(init[b], di[b], "numeric", True, "init$" + b)
            init[b] = np.asarray(init[b], dtype=np.float64).reshape(di[b])
        else:
            # matrix(runif(prod(dim)), ...): column-major fill
            init[b] = rng.random(di[b][0] * di[b][1]).reshape(di[b], order="F")
    return dict(
        Wm=np.concatenate([mask["W"], mask["W0"], mask["W1"]], axis=1),
        Hm=np.concatenate([mask["H"], mask["H1"], mask["H0"]], axis=0),
        Wi=np.concatenate([init["W"], init["W0"], init["W1"]], axis=1),
        Hi=np.concatenate([init["H"], init["H1"], init["H0"]], axis=0),
        kW0=kW0, kH0=kH0, K=K)


# ------------------------------------------------------------------------------------------------
# nnmf
# ------------------------------------------------------------------------------------------------
class NnmfResult(dict):
    """The reference's S3 object of class 'nnmf' (R/nnmf.R:184-224): a dict with attribute access."""

    __getattr__ = dict.__getitem__

    def __repr__(self):  # print.nnmf, R/nnmf_methods.R:53-80 (cosmetic, abbreviated)
        o = self["options"]
        return ("Non-negative matrix factorization:\n   Algorithm: %s\n        Loss: %s\n         MSE: %g\n         MKL: %g\n"
                "      Target: %g\n   Rel. tol.: %.3g\nTotal epochs: %d\n# Interation: %d\n" % (
                    {"scd": "Sequential coordinate-wise descent", "lee": "Lee's multiplicative algorithm"}[o["method"]],
                    {"mse": "Mean squared error", "mkl": "Mean Kullback-Leibler divergence"}[o["loss"]],
                    self["mse"][-1], self["mkl"][-1], self["target_loss"][-1],
                    abs(np.diff(self["target_loss"][-2:])[0] / np.mean(self["target_loss"][-2:])) if len(self["target_loss"]) > 1 else float("nan"),
                    int(np.sum(self["average_epochs"])), self["n_iteration"]))


def prepare_nnmf(A, k=1, alpha=(0, 0, 0), beta=(0, 0, 0), method="scd", loss="mse", init=None, mask=None, W_norm=-1,
                 check_k=True, max_iter=500, rel_tol=1e-4, n_threads=1, trace=None, verbose=1, show_warning=True,
                 inner_max_iter=None, inner_rel_tol=1e-9, rng=None):
    """Argument normalisation of nnmf(), R/nnmf.R:142-183 -> (17-tuple for c_nnmf, context dict)."""
    method = _match_arg(method, ("scd", "lee"), "method")
    loss = _match_arg(loss, ("mse", "mkl"), "loss")
    if inner_max_iter is None:
        inner_max_iter = 50 if loss == "mse" else 1  # R/nnmf.R:139
    if trace is None:
        trace = 100 / inner_max_iter  # R/nnmf.R:138
    A = np.asarray(A)
    if A.ndim != 2:
        raise NnlmStop("A must be a matrix")
    check_matrix(A, input_name="A")
    A = np.asarray(A, dtype=np.float64)
    n, m = A.shape
    im = reformat_input(init, mask, n, m, int(k), rng=rng)
    K = im["K"]
    alpha = np.concatenate([np.atleast_1d(np.asarray(alpha, dtype=np.float64)), np.zeros(3)])[:3]
    beta = np.concatenate([np.atleast_1d(np.asarray(beta, dtype=np.float64)), np.zeros(3)])[:3]
    code = get_method_code(method, loss)
    min_k = min(A.shape)
    isna = np.isnan(A)
    if isna.any():
        min_k = min(min_k, int((m - isna.sum(axis=1)).min()), int((n - isna.sum(axis=0)).min()))
    if check_k and K > min_k and np.all(np.concatenate([alpha, beta]) == 0):
        raise NnlmStop("k larger than %d is not recommended, unless properly masked or regularized.\n"
                       "\t\t\t\tSet check.k = FALSE if you want to skip this checking." % min_k)
    if n_threads < 0:
        n_threads = 0
    verbose = int(verbose)
    if trace <= 0:
        trace = 999999
    args = (A, int(K), im["Wi"], im["Hi"], im["Wm"], im["Hm"], alpha, beta, int(max_iter), float(rel_tol),
            int(n_threads), verbose, bool(show_warning), int(inner_max_iter), float(inner_rel_tol), code, int(trace))
    ctx = dict(method=method, loss=loss, alpha=alpha, beta=beta, init=init, mask=mask, n_threads=n_threads, trace=trace,
               verbose=verbose, max_iter=max_iter, rel_tol=rel_tol, inner_max_iter=inner_max_iter,
               inner_rel_tol=inner_rel_tol, W_norm=W_norm)
    return args, ctx


def finish_nnmf(out, ctx, run_time=None):
    """Result decoration of nnmf(), R/nnmf.R:184-224."""
    res = NnmfResult(W=np.array(out["W"]), H=np.array(out["H"]), mse=np.asarray(out["mse_error"]).ravel(),
                     mkl=np.asarray(out["mkl_error"]).ravel(), target_loss=np.asarray(out["target_error"]).ravel(),
                     average_epochs=np.asarray(out["average_epoch"]).ravel(), n_iteration=int(out["n_iteration"]))
    W_norm = ctx["W_norm"]
    if W_norm > 0:
        if np.isfinite(W_norm):
            scale = np.sum(res["W"] ** W_norm, axis=0) ** (1.0 / W_norm)
        else:
            scale = res["W"].max(axis=0)
        res["W"] = res["W"] @ np.diag(1.0 / scale)
        res["H"] = np.diag(scale) @ res["H"]
    res["run_time"] = run_time
    res["options"] = {key: ctx[key] for key in ("method", "loss", "alpha", "beta", "init", "mask", "n_threads", "trace",
                                                "verbose", "max_iter", "rel_tol", "inner_max_iter", "inner_rel_tol")}
    if out.get("warning"):
        warnings.warn("Target tolerance not reached. Try a larger max.iter.", RuntimeWarning, stacklevel=3)
    return res


def nnmf(A, k=1, alpha=(0, 0, 0), beta=(0, 0, 0), method="scd", loss="mse", init=None, mask=None, W_norm=-1,
         check_k=True, max_iter=500, rel_tol=1e-4, n_threads=1, trace=None, verbose=0, show_warning=True,
         inner_max_iter=None, inner_rel_tol=1e-9, rng=None):
    """Non-negative matrix factorisation A ~ W H on the MI355X (drop-in for R's NNLM::nnmf, R/nnmf.R:135-225).

    ``verbose`` defaults to 0 here (R: 1 = progress bar); ``rng`` seeds the default random init (R uses its global RNG).
    """
    args, ctx = prepare_nnmf(A, k, alpha, beta, method, loss, init, mask, W_norm, check_k, max_iter, rel_tol, n_threads,
                             trace, verbose, show_warning, inner_max_iter, inner_rel_tol, rng)
    g = rng or np.random.default_rng()
    cb = _lib.make_callbacks(unif_rand=lambda: g.random(), print_fn=(lambda s: print(s, end="")) if ctx["verbose"] == 2 else None)
    t0 = time.perf_counter()
    out = _lib.c_nnmf(*args, callbacks=cb)
    return finish_nnmf(out, ctx, run_time=time.perf_counter() - t0)


# ------------------------------------------------------------------------------------------------
# nnlm / predict
# ------------------------------------------------------------------------------------------------
class NnlmResult(dict):
    __getattr__ = dict.__getitem__


def _rcond(x):
    """R's rcond(x) for a tall matrix: reciprocal 1-norm condition number of the R factor of qr(x)."""
    r = np.linalg.qr(x, mode="r")
    try:
        return 1.0 / (np.linalg.norm(r, 1) * np.linalg.norm(np.linalg.inv(r), 1))
    except np.linalg.LinAlgError:
        return 0.0


def prepare_nnlm(x, y, alpha=(0, 0, 0), method="scd", loss="mse", init=None, mask=None, check_x=True, max_iter=10000,
                 rel_tol=1e-12, n_threads=1, show_warning=True):
    """Argument normalisation of nnlm(), R/nnlm.R:75-120 -> (9-tuple for c_nnlm, context)."""
    method = _match_arg(method, ("scd", "lee"), "method")
    loss = _match_arg(loss, ("mse", "mkl"), "loss")
    x = np.asarray(x)
    yv = np.asarray(y)
    with np.errstate(invalid="ignore"):
        if show_warning and loss == "mkl" and (np.any(x < 0) or np.any(yv < 0)):
            warnings.warn("x or y have negative values. One should instead use method == 'mse'.", RuntimeWarning, stacklevel=3)
    is_y_vector = yv.ndim == 1
    ym = _as_matrix(yv)
    check_matrix(ym, check_na=False)
    check_matrix(x, check_na=True)
    if x.ndim != 2:
        raise NnlmStop("x must be a matrix")
    if x.shape[0] != ym.shape[0]:
        raise NnlmStop("Dimensions of x and y do not match.")
    x = np.asarray(x, dtype=np.float64)
    ym = np.asarray(ym, dtype=np.float64)
    if max_iter <= 0:
        raise NnlmStop("max.iter must be positive.")
    if n_threads < 0:
        n_threads = 0
    if check_x:
        if x.shape[0] < x.shape[1] or _rcond(x) < np.finfo(np.float64).eps:
            warnings.warn("x does not have a full column rank. Solution may not be unique.", RuntimeWarning, stacklevel=3)
    alpha = np.concatenate([np.atleast_1d(np.asarray(alpha, dtype=np.float64)), np.zeros(3)])[:3]
    if show_warning and alpha[0] < alpha[1]:
        warnings.warn("If alpha[1] < alpha[2], be aware that that algorithm may not converge or unique.", RuntimeWarning, stacklevel=3)
    p, q = x.shape[1], ym.shape[1]
    if not _is_empty(mask):
        check_matrix(mask, dm=(p, q), mode="logical", check_na=True)
    if not _is_empty(init):
        check_matrix(init, dm=(p, q), check_na=True, check_negative=True)
    mask_m = None if _is_empty(mask) else np.asarray(mask, dtype=bool).reshape(p, q)
    init_m = None if _is_empty(init) else np.asarray(init, dtype=np.float64).reshape(p, q)
    if mask_m is not None and init_m is None:
        init_m = (~mask_m).astype(np.float64)  # masked entries fixed to 0, R/nnlm.R:110-112
    code = get_method_code(method, loss)
    args = (x, ym, alpha, mask_m, init_m, int(max_iter), float(rel_tol), int(n_threads), code)
    ctx = dict(method=method, loss=loss, max_iter=max_iter, rel_tol=rel_tol, is_y_vector=is_y_vector, alpha=alpha, x=x, y=ym)
    return args, ctx


def finish_nnlm(sol, ctx):
    """R/nnlm.R:122-144."""
    coef = np.array(sol["coefficient"])
    x, y, alpha, loss = ctx["x"], ctx["y"], ctx["alpha"], ctx["loss"]
    err = mse_mkl(y, x @ coef, na_rm=True, show_warning=False)
    target = 0.5 * err["MSE"] if loss == "mse" else err["MKL"]
    target = target + (alpha[0] - alpha[1]) * float(np.sum(coef ** 2)) + alpha[1] * float(np.sum(coef.sum(axis=0) ** 2)) \
        + alpha[2] * float(np.sum(coef))
    res = NnlmResult(coefficients=coef[:, 0] if ctx["is_y_vector"] else coef, n_iteration=int(sol["n_iteration"]),
                     error={"MSE": err["MSE"], "MKL": err["MKL"], "target.error": target},
                     options={"method": ctx["method"], "loss": loss, "max_iter": ctx["max_iter"], "rel_tol": ctx["rel_tol"]})
    return res


def nnlm(x, y, alpha=(0, 0, 0), method="scd", loss="mse", init=None, mask=None, check_x=True, max_iter=10000,
         rel_tol=1e-12, n_threads=1, show_warning=True, rng=None):
    """Non-negative linear model y ~ x beta on the MI355X (drop-in for R's NNLM::nnlm, R/nnlm.R:70-145)."""
    args, ctx = prepare_nnlm(x, y, alpha, method, loss, init, mask, check_x, max_iter, rel_tol, n_threads, show_warning)
    g = rng or np.random.default_rng()
    cb = _lib.make_callbacks(unif_rand=lambda: g.random())
    return finish_nnlm(_lib.c_nnlm(*args, callbacks=cb), ctx)


def predict_nnmf(object, newdata=None, which="A", method=None, loss=None, _nnlm=None, **kw):
    """S3 predict.nnmf, R/nnmf_methods.R:22-48.  ``_nnlm`` lets the CPU tests substitute the solver."""
    which = _match_arg(which, ("A", "W", "H"), "which")
    method = method or object["options"]["method"]
    loss = loss or object["options"]["loss"]
    solver = _nnlm or nnlm
    if which != "A":
        nd = np.asarray(newdata)
        if which == "W":
            check_matrix(nd, dm=(None, object["H"].shape[1]))
        if which == "H":
            check_matrix(nd, dm=(object["W"].shape[0], None))
        nd = np.asarray(nd, dtype=np.float64)
    if which == "A":
        return object["W"] @ object["H"]
    if which == "W":
        out = solver(object["H"].T, nd.T, method=method, loss=loss, **kw)
        out["coefficients"] = np.asarray(out["coefficients"]).T
        return out
    return solver(object["W"], nd, method=method, loss=loss, **kw)
