"""The benchmark's call (nnlm_run with trace = 2) without any profiling scope: for scripts/gpu_timeline.sh (TL_SCRIPT=gpu_run_trace.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nnlm_amd
from nnlm_amd import _lib

n, m, k = (int(v) for v in os.environ.get("SIZE", "20000,10000,50").split(","))
rng = np.random.default_rng(20250928)
A = rng.random((n, m)); W0 = 0.01 * rng.random((n, k)); H0 = 0.01 * rng.random((k, m))
z = [0, 0, 0]
with nnlm_amd.Handle(0, _lib.PREC_F32) as h:
    h.set_matrix(A); h.set_factors(k, W0, H0)
    r = h.run(z, z, int(os.environ.get("ITERS", "6")), -1.0, 0, False, 50, 1e-9, 1, 2)
    h.sync()
    print(r["mse_error"][-1])
