"""What makes the first timed region of bench.py slower than the later ones?  Same protocol (upload, 5 warm-up steps, 20 timed steps),
with nothing / an idle pause / a busy device in front of the warm-up."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import nnlm_amd
from nnlm_amd import _lib

n, m, k = 20000, 10000, 50
rng = np.random.default_rng(0)
A = rng.random((n, m)); W0 = rng.random((n, k)); H0 = rng.random((k, m))
z = [0.0, 0.0, 0.0]
def run(h, steps): return h.run(z, z, steps, -1.0, 0, False, 50, 1e-9, 1, 2)
for mode in ("plain", "idle 0.5 s", "busy 60 ms", "plain", "busy 60 ms", "busy 300 ms"):
    with nnlm_amd.Handle(0, _lib.PREC_F32) as h:
        h.set_matrix(A); h.set_factors(k, W0, H0)
        if mode.startswith("idle"): time.sleep(0.5)
        if mode.startswith("busy"):
            x = torch.randn(4096, 4096, device="cuda:0")
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < float(mode.split()[1]) * 1e-3:
                y = x @ x
                torch.cuda.synchronize()
        run(h, 5)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); run(h, 20); ts.append((time.perf_counter() - t0) / 20 * 1e3)
        print(mode, [round(t, 4) for t in ts], flush=True)
