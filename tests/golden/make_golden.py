"""Regenerates tests/golden/*.npz from the fp64 oracle (oracle/nnlm_ref.c via oracle/ref.py).

The reference itself cannot run here (no R), so these fixtures are outputs of OUR restatement,
which is pinned against the reference's own known-answer vectors (tests/golden/nnlm_kat.json,
from tests/testthat/test-nnlm.R) by tests/test_oracle.py.  Inputs come from numpy seeds so the
fixtures stay small: only seeds, shapes and outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref  # noqa: E402


def halfstep_inputs(seed, n, m, k, with_mask, with_na):
    rng = np.random.default_rng(seed)
    A = rng.random((n, m))
    Wt = rng.random((k, n))
    H = rng.random((k, m))
    mask = (rng.random((k, m)) < 0.15) if with_mask else None
    if with_mask:
        mask[:, 3] = True  # one fully masked column (skipped, src/update_with_missing.cpp:33)
    if with_na:
        A[rng.random((n, m)) < 0.1] = np.nan
        A[5, 7] = np.inf  # non-finite counts as missing too
    return A, Wt, H, mask


def driver_inputs(seed, n, m, k):
    rng = np.random.default_rng(seed)
    return rng.random((n, m)), 0.01 * rng.random((n, k)), 0.01 * rng.random((k, m))


def main():
    out = {}
    n, m, k = 40, 30, 5
    for method in (1, 2, 3, 4):
        for with_mask in (0, 1):
            for with_na in (0, 1):
                for reg_id, reg in enumerate(([0.0, 0.0, 0.0], [0.02, 0.01, 0.03])):
                    seed = 1000 + 100 * method + 10 * with_mask + with_na
                    A, Wt, H, mask = halfstep_inputs(seed, n, m, k, with_mask, with_na)
                    inner = 6 if method < 3 else 3
                    Hn, it = ref.update(H, Wt, A, mask, reg, inner, 1e-9, method)
                    key = f"hs_m{method}_k{with_mask}_na{with_na}_r{reg_id}"
                    out[key + "_H"] = Hn
                    out[key + "_it"] = np.int64(it)
                    out[key + "_meta"] = np.array([seed, n, m, k, inner], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "halfstep.npz"), **out)

    out = {}
    # config 1 of BASELINE.json: nnmf(A, k=5) MSE+SCD on 200 x 100, explicit init, fixed work (rel.tol=-1)
    A, W0, H0 = driver_inputs(20250928, 200, 100, 5)
    for name, method, inner, trace, iters, al, be in (
            ("cfg1_scd_mse", 1, 50, 2, 12, [0, 0, 0], [0, 0, 0]),
            ("cfg1_lee_mse", 2, 50, 2, 12, [0, 0, 0], [0, 0, 0]),
            ("cfg1_scd_mkl", 3, 1, 5, 12, [0, 0, 0], [0, 0, 0]),
            ("cfg1_lee_mkl", 4, 1, 5, 12, [0, 0, 0], [0, 0, 0]),
            ("cfg1_scd_mse_reg", 1, 50, 3, 10, [0.01, 0.0, 0.01], [0.02, 0.01, 0.0])):
        r = ref.c_nnmf(A, 5, W0, H0, None, None, al, be, iters, -1.0, 1, 0, True, inner, 1e-9, method, trace)
        for key in ("W", "H", "mse_error", "mkl_error", "target_error", "average_epoch"):
            out[f"{name}_{key}"] = r[key]
        out[f"{name}_n_iteration"] = np.int64(r["n_iteration"])
        out[f"{name}_args"] = np.array([method, inner, trace, iters] + al + be, dtype=np.float64)
    # config 5 shape (small): 10 % NA + L1/L2
    rng = np.random.default_rng(7)
    A5 = A.copy()
    A5.ravel()[rng.choice(A5.size, A5.size // 10, replace=False)] = np.nan
    r = ref.c_nnmf(A5, 5, W0, H0, None, None, [0.01, 0, 0.01], [0.01, 0, 0.01], 8, -1.0, 1, 0, True, 50, 1e-9, 1, 2)
    for key in ("W", "H", "mse_error", "mkl_error", "target_error", "average_epoch"):
        out[f"cfg5_na_{key}"] = r[key]
    out["cfg5_na_n_iteration"] = np.int64(r["n_iteration"])
    np.savez_compressed(os.path.join(HERE, "driver.npz"), **out)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
