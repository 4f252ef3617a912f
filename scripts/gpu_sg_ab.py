"""A/B of NNLM_SWEEP_GRAM (Gram partial sums from the sweep kernel's LDS image vs gram_partial_kernel): run once per setting
(the switch is read once per process), dump the factors, compare."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nnlm_amd
from nnlm_amd import _lib

def run(n, m, k, masked, iters, inner):
    rng = np.random.default_rng(77)
    A = rng.random((n, m)); W0, H0 = rng.random((n, k)), rng.random((k, m))
    Hm = (rng.random((k, m)) < 0.1) if masked else None
    reg = [0.02, 0.01, 0.03]
    with nnlm_amd.Handle(0, _lib.PREC_F32) as h:
        h.set_matrix(A); h.set_factors(k, W0, H0, None, Hm)
        h.iterate(iters, reg, reg, inner, 1e-9, 1)
        return h.get_factors()

if __name__ == "__main__":
    tag = sys.argv[1]
    out = {}
    for i, (n, m, k, masked, iters, inner) in enumerate([(300, 200, 9, True, 2, 6), (300, 200, 9, False, 1, 6), (300, 200, 9, False, 2, 6), (515, 131, 50, False, 3, 5), (1000, 900, 17, True, 4, 10)]):
        W, H = run(n, m, k, masked, iters, inner)
        out[f"W{i}"] = W; out[f"H{i}"] = H
    np.savez(f"/tmp/sg_{tag}.npz", **out)
    if tag == "1":
        a = np.load("/tmp/sg_0.npz")
        for i in range(5):
            print(i, "relF W %.2e H %.2e" % (np.linalg.norm(out[f"W{i}"] - a[f"W{i}"]) / np.linalg.norm(a[f"W{i}"]), np.linalg.norm(out[f"H{i}"] - a[f"H{i}"]) / np.linalg.norm(a[f"H{i}"])))
