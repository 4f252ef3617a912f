"""Probe (round 4): ONE kind of kernel per stream.  Handle X loops the cross-product phase of a sharded half-step (virtual rank 0 of 2,
reduce form: Gram + split copy + half of the A-streaming cross product + slab fold), handle S loops the sweep phase (its 10000 of the
20000 W columns, 50 sweeps) -- alone and concurrently.  The 96 KB two-stage cross product of the measurement in DESIGN.md section 6 (xprod16_tn_kernel<NKQ, 0, 2>, which leaves
room for a sweep workgroup on the same CU) was selected by a probe switch in launch_xprod16_m that has been removed again."""
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
    h.comm_init(None, 0, 2, form="reduce")
    h.set_matrix(A)
    h.set_factors(k, W0, H0)
    h.debug_phase(0, 1, z, 50, -1.0, 1)   # partial [G | C] of the W half-step in h->red: finite operands for the sweep phase
    h.sync()
X, S = hs
N = 30
def loop_x(cnt, which=1):
    for _ in range(cnt): X.debug_phase(which, 1, z, 50, -1.0, 1)
def loop_s(cnt):
    for _ in range(cnt): S.debug_phase(0, 2, z, 50, -1.0, 1)
def timed(fx, fs):
    for h in hs: h.sync()
    t0 = time.perf_counter()
    if fx: loop_x(fx)
    if fs: loop_s(fs)
    for h in hs: h.sync()
    return (time.perf_counter() - t0) * 1e3
loop_x(3); loop_s(3)
for rep in range(3):
    tx, ts, tb = timed(N, 0), timed(0, N), timed(N, N)
    print(f"rep {rep}: cross-product phase alone {tx / N:.4f} ms, sweep phase alone {ts / N:.4f} ms, both streams {tb / N:.4f} ms per pair "
          f"(sum {(tx + ts) / N:.4f}, max {max(tx, ts) / N:.4f})", flush=True)
for h in hs: h.close()
