"""Diagnosis aid for tests/test_gpu_fuzz.py (run on the GPU box: python tests/fuzz_table.py): one line per failing or degenerate case of the
first NNLM_FUZZ_SEEDS (default 150) seeds of both modes -- deviations of W and H, iteration and sweep counts, the case's parameters."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import test_gpu_fuzz as F
from helpers import relF
HARD = os.environ.get("NNLM_FUZZ_F32_HARD", "0") == "1"  # F32 mode on the strict mode's cases too (ranks up to the smaller dimension, up to 90 % missing)
for mode, wc in (("f64", False), ("f32", not HARD)):
    os.environ["NNLM_PRECISION"] = mode
    tol = 1e-9 if mode == "f64" else 1e-4
    for seed in range(int(os.environ.get("NNLM_FUZZ_SEEDS", "150"))):
        c = F.make_case(seed, wc)
        try:
            r, o = F.run_both(c)
        except Exception as e:
            print(mode, seed, "EXC", repr(e)[:200], F.describe(c)); continue
        ew, eh = relF(r["W"], o["W"]), relF(r["H"], o["H"])
        ep = np.array_equal(r["average_epoch"], o["average_epoch"]) if r["average_epoch"].shape == o["average_epoch"].shape else "shape"
        deg = F.degenerate(c)
        bad = not (ew < tol and eh < tol) or r["n_iteration"] != o["n_iteration"] or (mode == "f64" and ep is not True)
        if bad:
            d = F.describe(c)
            print(mode, seed, "DEG" if deg else "   ", f"W {ew:.2e} H {eh:.2e} nit {r['n_iteration']}/{o['n_iteration']} ep_eq {ep}", d["shape"], "k", d["k"], "meth", d["method"], f"na {d['na']:.2f}", "masks", d["masks"], "a", d["alpha"], "b", d["beta"], "it", d["max_iter"], "tr", d["trace"], "inner", d["inner"], flush=True)
