"""BASELINE.json configs[2] (KL + Lee) and configs[4] (10 % NA + L1/L2) at full size: timing + parity of one iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nnlm_amd
from nnlm_amd import _lib
from oracle import ref

def relF(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
n, m, k = (int(v) for v in os.environ.get("SIZE", "20000,10000,50").split(","))
check = os.environ.get("CHECK", "1") == "1"
rng = np.random.default_rng(20250928)
A = rng.random((n, m)); W0 = 0.01 * rng.random((n, k)); H0 = 0.01 * rng.random((k, m))
for name, method, inner, reg, na in (("cfg3 lee+mkl", 4, 1, [0, 0, 0], False), ("cfg3b scd+mkl", 3, 1, [0, 0, 0], False),
                                     ("cfg5 scd+mse NA+reg", 1, 50, [0.01, 0, 0.01], True)):
    A1 = A
    if na:
        A1 = A.copy(); A1.ravel()[np.random.default_rng(7).choice(n * m, n * m // 10, replace=False)] = np.nan
    with nnlm_amd.Handle(0, _lib.PREC_F32) as h:
        t0 = time.perf_counter(); h.set_matrix(A1); t1 = time.perf_counter()
        h.set_factors(k, W0, H0)
        h.iterate(1, reg, reg, inner, 1e-9, method); h.sync()
        W1, H1 = h.get_factors()
        its = 3
        per = []
        for _ in range(its):
            t2 = time.perf_counter(); h.iterate(1, reg, reg, inner, 1e-9, method); h.sync(); t3 = time.perf_counter()
            per.append(t3 - t2)
        t2, t3 = 0.0, sorted(per)[len(per) // 2] * its  # median iteration
        h.profile_enable(True)  # per-kernel times from a separate pass (HIP events around every scope)
        h.iterate(its, reg, reg, inner, 1e-9, method); h.sync()
        mse, kl, pen = h.errors()
        line = f"{name}: upload {t1-t0:.2f}s, {1e3*(t3-t2)/its:.2f} ms/iteration, mse {mse:.6f} kl {kl:.6f} |"
        for nm in ("xprod_h", "xprod_w", "gram", "sweep_h", "sweep_w", "errors"):
            ms, cnt = h.profile_get(nm); line += f" {nm} {ms/max(cnt,1):.3f}"
        print(line, flush=True)
    if check:
        t0 = time.perf_counter()
        Wt_ref, _ = ref.update(W0.T.copy(), H0, np.ascontiguousarray(A1.T), None, reg, inner, 1e-9, method)
        H_ref, _ = ref.update(H0, Wt_ref, A1, None, reg, inner, 1e-9, method)
        print(f"   oracle one iteration {time.perf_counter()-t0:.1f}s  relF(W) {relF(W1, Wt_ref.T):.2e} relF(H) {relF(H1, H_ref):.2e}", flush=True)
