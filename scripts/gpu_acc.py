"""Accuracy of SCD half-steps in the F32 mode against the oracle, split-fp16 vs fp32 cross products (run via gpurun:
NNLM_XPROD=f32 python scripts/gpu_acc.py ; python scripts/gpu_acc.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nnlm_amd
from nnlm_amd import _lib
from oracle import ref

def relF(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

out = []
for (n, m, k), inner, scale in (((515, 131, 50), 5, 1.0), ((64, 700, 64), 5, 1.0), ((2000, 1000, 50), 50, 1.0), ((2000, 1000, 50), 50, 1e5), ((2000, 1000, 50), 50, 1e-7)):
    rng = np.random.default_rng(n + m + k + 1)
    A = scale * rng.random((n, m)) ** 3          # wide dynamic range inside the matrix
    W0, H0 = rng.random((n, k)), np.sqrt(scale) * rng.random((k, m))
    reg = [0.02, 0.01, 0.03]
    with nnlm_amd.Handle(0, _lib.PREC_F32) as h:
        h.set_matrix(A); h.set_factors(k, W0, H0)
        h.half_step(0, reg, inner, 1e-9, 1)
        W1, _ = h.get_factors()
        h.half_step(1, reg, inner, 1e-9, 1)
        _, H1 = h.get_factors()
    Wt_ref, _ = ref.update(W0.T.copy(), H0, A.T.copy(), None, reg, inner, 1e-9, 1)
    H_ref, _ = ref.update(H0.copy(), Wt_ref, A, None, reg, inner, 1e-9, 1)
    out.append(f"{n}x{m} k={k} inner={inner} scale={scale:g}: W {relF(W1, Wt_ref.T):.2e} H {relF(H1, H_ref):.2e}")
print(f"NNLM_XPROD={os.environ.get('NNLM_XPROD', 'f16x2 (default)')} | " + " | ".join(out), flush=True)
