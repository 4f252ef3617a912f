"""Parity of the fp32-operand mode at the depth it is advertised at ("next round" 5).

Config 2 (20000 x 10000, k = 50, MSE + SCD, trace 2, inner.max.iter 50, rel.tol -1) for 200 iterations on the GPU and with
oracle/nnlm_ref.c on the box's host threads, from the same explicit init; relF(W), relF(H) and the largest trace difference at
iterations 20 / 50 / 100 / 200.  Then the R defaults (R/nnmf.R:138-140: max.iter = 500, rel.tol = 1e-4): n.iteration of both.
Test infrastructure (imports oracle/): run on the GPU box, writes gpurun_out/r06_parity_depth.json (copy to profiles/).

    python scripts/gpu_parity_depth.py [--stages 20,50,100,200] [--defaults 1] [--size n,m,k]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nnlm_amd  # noqa: E402
from nnlm_amd import _lib  # noqa: E402
from oracle import ref  # noqa: E402


def relF(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stages", default="20,50,100,200")
    ap.add_argument("--defaults", type=int, default=1)
    ap.add_argument("--size", default="20000,10000,50")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_parity_depth.json"))
    args = ap.parse_args()
    n, m, k = (int(v) for v in args.size.split(","))
    stages = [int(v) for v in args.stages.split(",")]
    rng = np.random.default_rng(20250928)  # SURVEY section 8d's generator
    A = np.asfortranarray(rng.random((n, m)))
    W0, H0 = 0.01 * rng.random((n, k)), 0.01 * rng.random((k, m))
    z = [0.0, 0.0, 0.0]
    threads = os.cpu_count() or 1
    out = dict(config=dict(n=n, m=m, k=k, method=1, inner_max_iter=50, inner_rel_tol=1e-9, trace=2, rel_tol=-1), host_threads=threads, stages=[])

    # ---- fixed iteration counts: both sides advance stage by stage from their OWN factors (same operations as one long run: the stage
    # lengths are even, so every trace iteration falls where it would)
    h = nnlm_amd.Handle(0, _lib.PREC_F32)
    h.set_matrix(A)
    h.set_factors(k, W0, H0)
    Wo, Ho = W0, H0
    done = 0
    for upto in stages:
        it = upto - done
        t0 = time.perf_counter()
        tg = h.run(z, z, it, -1.0, 0, False, 50, 1e-9, 1, 2)
        h.sync()
        gpu_s = time.perf_counter() - t0
        Wg, Hg = h.get_factors()
        t0 = time.perf_counter()
        o = ref.c_nnmf(A, k, Wo, Ho, None, None, z, z, it, -1.0, threads, 0, False, 50, 1e-9, 1, 2)
        cpu_s = time.perf_counter() - t0
        Wo, Ho = o["W"], o["H"]
        done = upto
        ntr = min(len(tg["mse_error"]), len(o["mse_error"]))
        rec = dict(iterations=upto, relF_W=relF(Wg, Wo), relF_H=relF(Hg, Ho),
                   max_rel_mse_trace=float(np.max(np.abs(tg["mse_error"][:ntr] - o["mse_error"][:ntr]) / o["mse_error"][:ntr])),
                   max_rel_target_trace=float(np.max(np.abs(tg["target_error"][:ntr] - o["target_error"][:ntr]) / o["target_error"][:ntr])),
                   max_abs_epoch_trace=float(np.max(np.abs(tg["average_epoch"][:ntr] - o["average_epoch"][:ntr]))),
                   mse_gpu=float(tg["mse_error"][-1]), mse_oracle=float(o["mse_error"][-1]), gpu_seconds=gpu_s, oracle_seconds=cpu_s)
        out["stages"].append(rec)
        print(json.dumps(rec), flush=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
    h.close()

    # ---- the R defaults: max.iter = 500, rel.tol = 1e-4 (R/nnmf.R:138-140); n.iteration of both
    if args.defaults:
        with nnlm_amd.Handle(0, _lib.PREC_F32) as h2:
            h2.set_matrix(A)
            h2.set_factors(k, W0, H0)
            t0 = time.perf_counter()
            tg = h2.run(z, z, 500, 1e-4, 0, True, 50, 1e-9, 1, 2)
            h2.sync()
            gpu_s = time.perf_counter() - t0
            Wg, Hg = h2.get_factors()
        t0 = time.perf_counter()
        o = ref.c_nnmf(A, k, W0, H0, None, None, z, z, 500, 1e-4, threads, 0, True, 50, 1e-9, 1, 2)
        cpu_s = time.perf_counter() - t0
        rec = dict(max_iter=500, rel_tol=1e-4, n_iteration_gpu=int(tg["n_iteration"]), n_iteration_oracle=int(o["n_iteration"]),
                   warned_gpu=bool(tg["warning"]), warned_oracle=bool(o["warning"]), relF_W=relF(Wg, o["W"]), relF_H=relF(Hg, o["H"]),
                   mse_gpu=float(tg["mse_error"][-1]), mse_oracle=float(o["mse_error"][-1]), gpu_seconds=gpu_s, oracle_seconds=cpu_s)
        out["r_defaults"] = rec
        print(json.dumps(rec), flush=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
