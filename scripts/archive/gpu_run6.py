"""Six resident iterations with trace = 2 (the bench protocol) for timeline inspection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nnlm_amd
from nnlm_amd import _lib
n, m, k = 20000, 10000, 50
rng = np.random.default_rng(20250928)
A = rng.random((n, m)); W0 = 0.01 * rng.random((n, k)); H0 = 0.01 * rng.random((k, m))
z = [0, 0, 0]
with nnlm_amd.Handle(0, _lib.PREC_F32) as h:
    h.set_matrix(A); h.set_factors(k, W0, H0)
    h.run(z, z, 4, -1.0, 0, False, 50, 1e-9, 1, 2)
    h.sync()
    r = h.run(z, z, 6, -1.0, 0, False, 50, 1e-9, 1, 2)
    print(r["mse_error"])
