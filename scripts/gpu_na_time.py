"""BASELINE.json configs[4] (10 % NA + L1/L2) at full size: wall time of successive iterations (no profiling scopes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    out = []
    for it in range(8):
        t2 = time.perf_counter(); h.iterate(1, reg, reg, 50, 1e-9, 1); h.sync(); t3 = time.perf_counter()
        out.append(f"{1e3 * (t3 - t2):.1f}")
    print("ms per iteration, iterations 1..8:", " ".join(out), "| sweeps per column so far", h.take_sweeps() / (n + m) / 8, flush=True)
