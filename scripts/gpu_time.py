"""Per-kernel timing at BASELINE.json configs[1] through the library's HIP-event profiler (run via gpurun)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nnlm_amd
from nnlm_amd import _lib

n, m, k = (int(v) for v in os.environ.get("SIZE", "20000,10000,50").split(","))
prec = _lib.PREC_F64 if os.environ.get("PREC", "f32") == "f64" else _lib.PREC_F32
iters = int(os.environ.get("ITERS", "10"))
inner = int(os.environ.get("INNER", "50"))
rng = np.random.default_rng(20250928)
A = rng.random((n, m)); W0 = 0.01 * rng.random((n, k)); H0 = 0.01 * rng.random((k, m))
z = [0, 0, 0]
with nnlm_amd.Handle(0, prec) as h:
    h.set_matrix(A); h.set_factors(k, W0, H0)
    h.iterate(2, z, z, inner, 1e-9, 1); h.errors(); h.sync()
    h.take_sweeps()
    h.profile_enable(True)
    t0 = time.perf_counter(); h.iterate(iters, z, z, inner, 1e-9, 1); h.sync(); dt = time.perf_counter() - t0
    mse = h.errors()[0]
    sw = h.take_sweeps() / (n + m) / iters
    line = f"inner={inner} {iters} it {1e3*dt/iters:.3f} ms/it sweeps/col {sw:.2f} mse {mse:.9f} |"
    for nm in ("xprod_h", "xprod_w", "gram", "sweep_h", "sweep_w", "errors"):
        ms, cnt = h.profile_get(nm); line += f" {nm} {ms/max(cnt,1):.3f}"
    print(line, flush=True)
