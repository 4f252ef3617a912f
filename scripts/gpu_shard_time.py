"""Per-rank kernel times of the sharded dense half-step at config 2, both forms (column-sharded: phase "contract" is empty and "sweep" is
cross product + Gram + sweep of the rank's columns; reduce: contraction slab, then sweep), measured on ONE GPU with virtual ranks (the
collectives are not timed: a virtual rank has no communicator).  Feeds DESIGN.md section 6's scaling model."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nnlm_amd
from nnlm_amd import _lib
n, m, k = 20000, 10000, 50
rng = np.random.default_rng(20250928)
A = rng.random((n, m)); W0 = 0.01 * rng.random((n, k)); H0 = 0.01 * rng.random((k, m))
z = [0.0, 0.0, 0.0]
out = {}
for form, world in [("cols", 1)] + [(f, w) for f in ("cols", "reduce") for w in (2, 4, 8)]:
    with nnlm_amd.Handle(0, _lib.PREC_F32) as h:
        if world > 1:
            h.comm_init(None, world - 1, world, form=form)  # the last rank (ragged shard)
        h.set_matrix(A); h.set_factors(k, W0, H0)
        res = {}
        for which in (0, 1):
            ts = {1: [], 2: [], 3: []}
            # inner.rel.tol = -1: always 50 sweeps (the benchmark's regime); the unpack is timed last (a virtual rank's gathered
            # buffer was never filled, so it would overwrite the factor with garbage)
            # (round 3: the unpack of a half-step leaves the Gram for the next one, so the three phases alternate -- a virtual rank's
            #  gathered buffer was never filled: the factors are garbage after the first unpack, which the forced 50 sweeps do not mind;
            #  the factors are reset after the loop)
            for rep in range(6):
                if world == 1:
                    h.sync(); t0 = time.perf_counter(); h.half_step(which, z, 50, -1.0, 1); h.sync(); ts[1].append(time.perf_counter() - t0)
                else:
                    for ph in (1, 2, 3):
                        h.sync(); t0 = time.perf_counter(); h.debug_phase(which, ph, z, 50, -1.0, 1); h.sync(); ts[ph].append(time.perf_counter() - t0)
                    # the OTHER half-step's sweep phase finds this one's Gram: time it too
                    h.sync(); t0 = time.perf_counter(); h.debug_phase(1 - which, 2, z, 50, -1.0, 1); h.sync(); ts.setdefault(4, []).append(time.perf_counter() - t0)
            if world > 1:
                h.set_factors(k, W0, H0)
            res["W" if which == 0 else "H"] = {("half_step" if world == 1 else {1: "contract", 2: "sweep", 3: "unpack", 4: "next_sweep_with_gathered_gram"}[ph]): round(1e3 * min(v), 4) for ph, v in ts.items() if v}
        out[f"{form}_{world}" if world > 1 else "1"] = res
        print(form, world, json.dumps(res), flush=True)
os.makedirs("gpurun_out/r04", exist_ok=True)
json.dump(out, open("gpurun_out/r04/shard_times.json", "w"))
