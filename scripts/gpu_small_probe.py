"""Small problems: wall time per outer iteration of nnlm_c_nnmf (both modes) against the oracle's C restatement on one host thread."""
import sys, time, os
import numpy as np
sys.path.insert(0, ".")
import nnlm_amd
from oracle import ref

rng = np.random.default_rng(0)
z = [0.0, 0.0, 0.0]
for (n, m, k) in [(200, 100, 5), (1000, 500, 10), (2000, 1000, 20), (5000, 2000, 30)]:
    A = rng.random((n, m)); W0 = rng.random((n, k)); H0 = rng.random((k, m))
    its = 100
    for mode in ("f32", "f64"):
        os.environ["NNLM_PRECISION"] = mode
        nnlm_amd.c_nnmf(A, k, W0, H0, None, None, z, z, 4, -1.0, 1, 0, False, 50, 1e-9, 1, 2)  # warm
        t0 = time.perf_counter(); r = nnlm_amd.c_nnmf(A, k, W0, H0, None, None, z, z, its, -1.0, 1, 0, False, 50, 1e-9, 1, 2); t1 = time.perf_counter()
        print(n, m, k, mode, "GPU ms/it", round((t1 - t0) / its * 1e3, 4), "call s", round(t1 - t0, 4), flush=True)
    ci = 5 if n * m > 1e6 else 20
    t0 = time.perf_counter(); ref.c_nnmf(A, k, W0, H0, None, None, z, z, ci, -1.0, 1, 0, False, 50, 1e-9, 1, 2); t1 = time.perf_counter()
    print(n, m, k, "oracle (1 thread) ms/it", round((t1 - t0) / ci * 1e3, 3), flush=True)
