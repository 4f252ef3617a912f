import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nnlm_amd
from nnlm_amd import _lib
from oracle import ref
def relF(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
n, m = 300, 170
for k in (40, 64, 65, 80):
    for method in (3, 4):
        for zeros in (False, True):
            for inner in (1, 2):
                rng = np.random.default_rng(k + method)
                A = rng.random((n, m)); W0, H0 = rng.random((n, k)), rng.random((k, m))
                if zeros:
                    Hm = rng.random((k, m)) < 0.1; Hm[:, 3] = True; H0[Hm] = 0.0
                reg = [0.02, 0.01, 0.03]
                Wt_ref, it1 = ref.update(W0.T.copy(), H0, A.T.copy(), None, reg, inner, 1e-9, method)
                line = f"k={k} m{method} zeros={int(zeros)} inner={inner}:"
                for pname, prec in (("f64", _lib.PREC_F64), ("f32", _lib.PREC_F32)):
                    with nnlm_amd.Handle(0, prec) as h:
                        h.set_matrix(A); h.set_factors(k, W0, H0)
                        h.half_step(0, reg, inner, 1e-9, method)
                        W1, _ = h.get_factors()
                    d = np.abs(W1 - Wt_ref.T)
                    bad = np.where(d.max(axis=1) > 1e-3)[0]
                    line += f"  {pname} relF {relF(W1, Wt_ref.T):.2e} badcols {len(bad)} {bad[:6]}"
                print(line, flush=True)
