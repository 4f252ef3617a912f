"""Time of nnlm_set_matrix (upload through the staging pipeline + prep pass + resident copies) at the benchmark size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnlm_amd
from nnlm_amd import _lib
n, m = 20000, 10000
A = np.asfortranarray(np.random.default_rng(1).random((n, m)))
for name, prec in (("f32", _lib.PREC_F32), ("f64", _lib.PREC_F64)):
    with nnlm_amd.Handle(0, prec) as h:
        for rep in range(3):
            t0 = time.perf_counter()
            h._ck(h._lib.nnlm_set_matrix(h._h, A.ctypes.data_as(_lib.C.POINTER(_lib.C.c_double)), n, m))
            dt = time.perf_counter() - t0
            print(f"{name} set_matrix {dt:.3f} s = {A.nbytes / dt / 1e9:.1f} GB/s of the caller's fp64 matrix; info {h.matrix_info() if rep == 0 else ''}", flush=True)
