"""Where the wall time of a cold nnlm_c_nnmf() call goes (SURVEY 8d's call-level metric): the resident API's steps timed one by one in a
fresh process -- create (HIP runtime + code object), set_matrix (allocations, upload, conversions), set_factors, run, get_factors,
destroy -- then the same sequence again (warm)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t_imp = time.perf_counter()
import nnlm_amd
from nnlm_amd import _lib
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
n, m, k = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (20000, 10000, 50)
its = int(sys.argv[5]) if len(sys.argv) > 5 else 200
rng = np.random.default_rng(20250928)
A = np.asfortranarray(rng.random((n, m))); W0 = 0.01 * rng.random((n, k)); H0 = 0.01 * rng.random((k, m))
z = [0.0, 0.0, 0.0]
out = {}
for rnd in ("cold", "warm", "warm2"):
    t = {}
    t0 = time.perf_counter(); h = nnlm_amd.Handle(0, _lib.PREC_F64 if prec == "f64" else _lib.PREC_F32); t["create"] = time.perf_counter() - t0
    t0 = time.perf_counter(); h.set_matrix(A); t["set_matrix"] = time.perf_counter() - t0
    t0 = time.perf_counter(); h.set_factors(k, W0, H0); t["set_factors"] = time.perf_counter() - t0
    t0 = time.perf_counter(); r = h.run(z, z, its, -1.0, 0, False, 50, 1e-9, 1, 2); t["run_200" if its == 200 else f"run_{its}"] = time.perf_counter() - t0
    t0 = time.perf_counter(); h.get_factors(); t["get_factors"] = time.perf_counter() - t0
    t0 = time.perf_counter(); h.close(); t["destroy"] = time.perf_counter() - t0
    t["total"] = sum(t.values())
    out[rnd] = {kk: round(v, 4) for kk, v in t.items()}
print(json.dumps(dict(precision=prec, **out)))
