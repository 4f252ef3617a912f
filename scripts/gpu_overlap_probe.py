"""Probe (round 4): do the HBM-bound cross products and the latency-bound sweeps of TWO independent nnmf problems overlap when their
half-steps are enqueued on different streams of one GPU?  Two resident handles (each owns its streams) with the same config-2
problem; `iterate` only enqueues.  t_pair close to t_single = full overlap, 2 x t_single = none."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnlm_amd
from nnlm_amd import _lib

n, m, k = 20000, 10000, 50
rng = np.random.default_rng(20250928)
A = rng.random((n, m))
W0, H0 = 0.01 * rng.random((n, k)), 0.01 * rng.random((k, m))
z = [0.0, 0.0, 0.0]
hs = [nnlm_amd.Handle(0, _lib.PREC_F32) for _ in range(2)]
for h in hs:
    h.set_matrix(A)
    h.set_factors(k, W0, H0)
    h.iterate(3, z, z, 50, 1e-9, 1)
    h.sync()
IT = 20
def timed(handles):
    for h in handles: h.sync()
    t0 = time.perf_counter()
    for h in handles: h.iterate(IT, z, z, 50, 1e-9, 1)
    for h in handles: h.sync()
    return (time.perf_counter() - t0) / IT * 1e3
for rep in range(3):
    a = timed(hs[:1]); b = timed(hs[1:]); c = timed(hs)
    print(f"rep {rep}: single {a:.4f} / {b:.4f} ms per iteration; two problems concurrently {c:.4f} ms per iteration-pair  (ratio {c / (0.5 * (a + b)):.3f})", flush=True)
# stagger: the second handle starts half an iteration later (its cross products meet the other's sweeps)
for h in hs: h.sync()
t0 = time.perf_counter()
hs[0].half_step(0, z, 50, 1e-9, 1)
for i in range(IT):
    hs[1].half_step(0, z, 50, 1e-9, 1); hs[0].half_step(1, z, 50, 1e-9, 1)
    hs[1].half_step(1, z, 50, 1e-9, 1); hs[0].half_step(0, z, 50, 1e-9, 1)
for h in hs: h.sync()
print(f"staggered enqueue: {(time.perf_counter() - t0) / IT * 1e3:.4f} ms per iteration-pair", flush=True)
for h in hs: h.close()
