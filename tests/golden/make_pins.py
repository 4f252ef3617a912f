"""Generates tests/golden/pins.json: reference-INDEPENDENT answers for the solvers the reference holds no vectors for.

Neither oracle restatement is involved in producing these numbers:
  * least squares (scd_ls_update, lee_ls_update, update_with_missing): the non-negative least-squares problems the solvers
    iterate on, solved by scipy.optimize.nnls (Lawson-Hanson active set) -- the way the reference made its own known answer
    (tests/testthat/test-nnlm.R:39 used nnls::nnls).  Penalties enter through an augmented system:
        min 1/2 |W x - y|^2 + 1/2 (b0 - b1) |x|^2 + 1/2 b1 (sum x)^2 + b2 sum x,  x >= 0
      = NNLS on [W; sqrt(b0 - b1) I; sqrt(b1) 1^T] with the right-hand side shifted so that its normal equations carry -b2;
    missing values: the rows of a column's problem are the finite entries of that column (src/update_with_missing.cpp:86-111).
  * KL (scd_kl_update, lee_kl_update): min_h sum_i (W h)_i - a_i log((W h)_i) [+ penalties], h >= 0, solved by
    scipy.optimize.minimize(L-BFGS-B, analytic gradient, bounds) from several starts.
Run:  python tests/golden/make_pins.py   (scipy is test-time only; the product never imports it)
"""
import json
import os

import numpy as np
from scipy.optimize import minimize, nnls

HERE = os.path.dirname(os.path.abspath(__file__))


def ls_problem(rng, n, k, cond, nneg):
    """Design with prescribed condition number; truth with `nneg` coordinates that want to be negative (-> active at 0)."""
    U, _ = np.linalg.qr(rng.standard_normal((n, k)))
    V, _ = np.linalg.qr(rng.standard_normal((k, k)))
    s = np.logspace(0, -np.log10(cond), k)
    W = np.abs(U * s @ V.T) + 0.05 * rng.random((n, k))  # non-negative design, like a factor
    x = rng.random(k)
    x[rng.permutation(k)[:nneg]] *= -1.0
    y = W @ x + 0.01 * rng.standard_normal(n)
    return W, np.abs(y)


def nnls_pen(W, y, b):
    n, k = W.shape
    b0, b1, b2 = b
    rows = [W]
    if b0 != b1:
        rows.append(np.sqrt(b0 - b1) * np.eye(k))
    if b1 != 0:
        rows.append(np.sqrt(b1) * np.ones((1, k)))
    Wa = np.vstack(rows)
    ya = np.concatenate([y, np.zeros(Wa.shape[0] - n)])
    if b2 != 0:
        ya = ya - b2 * Wa @ np.linalg.solve(Wa.T @ Wa, np.ones(k))
    x, _ = nnls(Wa, ya, maxiter=100 * k)
    return x


def kl_min(W, a, b, starts, rng):
    n, k = W.shape
    b0, b1, b2 = b

    def f(h):
        wh = W @ h + 1e-16
        s = h.sum()
        val = np.sum(wh - a * np.log(wh)) + 0.5 * (b0 - b1) * h @ h + 0.5 * b1 * s * s + b2 * s
        g = W.T @ (1.0 - a / wh) + (b0 - b1) * h + b1 * s + b2
        return val, g

    best = None
    for _ in range(starts):
        r = minimize(f, rng.random(k) + 0.1, jac=True, method="L-BFGS-B", bounds=[(0, None)] * k,
                     options=dict(maxiter=20000, ftol=1e-15, gtol=1e-12, maxcor=30))
        if best is None or r.fun < best.fun:
            best = r
    return best.x, float(best.fun)


def main():
    rng = np.random.default_rng(20250929)
    out = {"ls": [], "ls_na": [], "kl": []}
    for i in range(20):
        n, k = int(rng.integers(12, 40)), int(rng.integers(3, 9))
        cond = [1e1, 1e2, 1e3, 1e4][i % 4]
        W, y = ls_problem(rng, n, k, cond, nneg=i % 3)
        b = [[0, 0, 0], [0.05, 0, 0], [0.05, 0.02, 0], [0.05, 0.02, 0.03], [0, 0, 0.04]][i % 5]
        out["ls"].append(dict(W=W.tolist(), y=y.tolist(), beta=b, cond=cond, x=nnls_pen(W, y, b).tolist()))
    for i in range(6):  # several columns with their own missing rows
        n, k, m = 30, 4, 5
        W, _ = ls_problem(rng, n, k, 1e2, 0)
        A = np.abs(W @ rng.random((k, m)) + 0.05 * rng.standard_normal((n, m)))
        miss = rng.random((n, m)) < 0.2
        b = [[0, 0, 0], [0.03, 0.01, 0.02]][i % 2]
        X = np.stack([nnls_pen(W[~miss[:, j]], A[~miss[:, j], j], b) for j in range(m)], axis=1)
        out["ls_na"].append(dict(W=W.tolist(), A=np.where(miss, np.nan, A).tolist(), beta=b, X=X.tolist()))
    for i in range(10):
        n, k = int(rng.integers(15, 40)), int(rng.integers(2, 6))
        W = rng.random((n, k)) + 0.05
        h = rng.random(k)
        if i % 3 == 1:
            h[0] = 0.0
        a = rng.poisson(20 * W @ h).astype(float) / 20 + (0.0 if i % 2 else 0.05)
        b = [[0, 0, 0], [0.05, 0.02, 0.03]][i % 2]
        x, fun = kl_min(W, a, b, 4, rng)
        # scd_kl_update puts beta(0) into the curvature only (src/base_algorithms.cpp:98-100: the gradient it uses is
        # beta(2) + beta(1) * (sum(Hj) - Hj(k)), without beta(0) * Hj(k)), so its fixed point minimises the objective with b0 = 0
        xs, funs = kl_min(W, a, [0.0, b[1], b[2]], 4, rng)
        out["kl"].append(dict(W=W.tolist(), a=a.tolist(), beta=b, x=x.tolist(), fun=fun, x_scd=xs.tolist(), fun_scd=funs))
    with open(os.path.join(HERE, "pins.json"), "w") as fh:
        json.dump(out, fh)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
