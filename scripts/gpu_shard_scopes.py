"""Profile scopes (HIP events around each kernel class) of ONE virtual rank's column-sharded dense half-step at config 2."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nnlm_amd
from nnlm_amd import _lib
n, m, k = 20000, 10000, 50
rng = np.random.default_rng(20250928)
A = rng.random((n, m)); W0 = 0.01 * rng.random((n, k)); H0 = 0.01 * rng.random((k, m))
z = [0.0, 0.0, 0.0]
for world in (8, 4, 2):
    with nnlm_amd.Handle(0, _lib.PREC_F32) as h:
        h.comm_init(None, world - 1, world)
        h.set_matrix(A); h.set_factors(k, W0, H0)
        for which in (0, 1):
            for _ in range(2):
                h.debug_phase(which, 2, z, 50, -1.0, 1)
            h.sync(); h.profile_enable(True); h.profile_reset()
            for _ in range(5):
                h.debug_phase(which, 2, z, 50, -1.0, 1)
            h.sync()
            out = {}
            for nm in ("xprod_h", "xprod_w", "gram", "sweep_h", "sweep_w"):
                t, c = h.profile_get(nm)
                if c: out[nm] = round(t / c, 4)
            h.profile_enable(False)
            print(world, "W" if which == 0 else "H", json.dumps(out), flush=True)
