import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import nnlm_amd
from nnlm_amd import _lib
n, m, k = 20000, 10000, 50
rng = np.random.default_rng(20250928)
A = rng.random((n, m)); W0 = 0.01 * rng.random((n, k)); H0 = 0.01 * rng.random((k, m))
A.ravel()[np.random.default_rng(7).choice(n * m, n * m // 10, replace=False)] = np.nan
reg = [0.01, 0, 0.01]
with nnlm_amd.Handle(0, _lib.PREC_F32) as h:
    h.set_matrix(A); h.set_factors(k, W0, H0)
    h.iterate(1, reg, reg, 50, 1e-9, 1); h.sync()
    for its in (1, 3, 10):
        t2 = time.perf_counter(); h.iterate(its, reg, reg, 50, 1e-9, 1); h.sync(); t3 = time.perf_counter()
        print(its, "iterations:", round(1e3 * (t3 - t2) / its, 2), "ms/iteration", flush=True)
