"""Time the stand-alone error block (errors_f32_kernel; `gpu_err_time.py f64`: errors64_kernel) at BASELINE sizes, dense and with 10 %
missing entries, and check its two sums against a host evaluation in fp64 (GPU box only)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnlm_amd  # noqa: E402
from nnlm_amd import _lib  # noqa: E402

N, M, K = 20000, 10000, 50
PREC = _lib.PREC_F64 if "f64" in sys.argv[1:] else _lib.PREC_F32
rng = np.random.default_rng(1)
A = rng.random((N, M))
W, H = rng.random((N, K)) * 0.2, rng.random((K, M)) * 0.2
for tag in ("dense", "na"):
    if tag == "na":
        A = A.copy()
        A.ravel()[np.random.default_rng(7).choice(N * M, N * M // 10, replace=False)] = np.nan
    with nnlm_amd.Handle(0, PREC) as h:
        h.set_matrix(A)
        h.set_factors(K, W, H)
        mse, kl, _ = h.errors()
        klc = h.matrix_info()["kl_const"]
        h.profile_enable(True)
        h.profile_reset()
        t0 = time.perf_counter()
        for _ in range(10):
            h.errors()
        dt = (time.perf_counter() - t0) / 10
        tot, cnt = h.profile_get("errors")
    ok = np.isfinite(A)
    Ah = W @ H
    ref_mse = float(np.mean((np.where(ok, A, 0) - Ah)[ok] ** 2))
    ref_kl = float(np.mean((-(np.where(ok, A, 1) + 1e-16) * np.log(Ah + 1e-16) + Ah)[ok]))
    print(f"{tag}: errors scope {tot / cnt:.3f} ms/launch ({cnt} launches), wall {1e3 * dt:.3f} ms/call; mse rel diff {abs(mse - ref_mse) / ref_mse:.2e}, "
          f"kl-var rel diff {abs(kl - ref_kl) / abs(ref_kl):.2e}", flush=True)
