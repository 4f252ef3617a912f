#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on N GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one outer iteration of nnmf(): W half-step + H half-step (src/nnmf.cpp:114-133) plus, every `trace`
steps, the error block (src/nnmf.cpp:135-160).  Workload = BASELINE.json configs[1]: nnmf(A, k=50), MSE loss,
sequential coordinate descent, dense A 20000 x 10000 = U(0,1) synthetic, explicit 0.01*U(0,1) init, R defaults
inner.max.iter=50, inner.rel.tol=1e-9, trace=2, rel.tol=-1 (fixed work).  A, W, H are resident in HBM when the timed
region starts.  N > 1: A replicated, each rank contracts its slab, one RCCL all-reduce per half-step (strong scaling).

Prints ONE JSON line on rank 0 (see the task contract) with `roofline` (the A-streaming cross-product kernel,
HIP-event timed inside this process) and `cpu_baseline` (oracle/nnlm_ref.c, OpenMP, on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_, M_, K_ = 20000, 10000, 50
INNER, INNER_TOL, METHOD = 50, float(os.environ.get("NNLM_BENCH_INNER_TOL", "1e-9")), 1  # (env: experiments only)
SEED = 20250928


def make_inputs(n, m, k):
    rng = np.random.default_rng(SEED)
    A = rng.random((n, m))
    W0 = 0.01 * rng.random((n, k))
    H0 = 0.01 * rng.random((k, m))
    return A, W0, H0


def run_steps(h, steps, trace, first_index=0):
    """`steps` outer iterations of the reference loop (src/nnmf.cpp:109-161) on the resident problem: W half-step,
    H half-step, error block every `trace` iterations (+ the closing one), rel.tol = -1 so that no iteration is skipped.
    This is nnlm_run(), the same code path nnlm_c_nnmf() takes after its upload.  Returns the last mse."""
    z = [0.0, 0.0, 0.0]
    r = h.run(z, z, steps, -1.0, 0, False, INNER, INNER_TOL, METHOD, trace if trace > 0 else 999999)
    assert r["n_iteration"] == steps
    return float(r["mse_error"][-1])


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/*_pmc_hbm_summary.txt,
    separate --pmc FETCH_SIZE / WRITE_SIZE passes over this same command; KiB per dispatch).  gfx950 correction
    (MI355X_MICROARCH.md, HBM section): a wide streaming read is tallied at half its bytes, so reads = 2 x FETCH_SIZE.
    bench.py cannot attach rocprofv3 to itself: (None, None) when no summary is committed."""
    import glob, re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_hbm_summary.txt")),
                   key=lambda f: [int(t) for t in re.findall(r"\d+", os.path.basename(f))])  # r01_v10 after r01_v9
    if not files:
        return None, None
    fetch = write = None
    for line in open(files[-1]):
        if kernel in line:
            mt = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+\s+mean=([0-9.eE+-]+)", line)
            if mt and mt.group(1) == "FETCH_SIZE":
                fetch = float(mt.group(2))
            elif mt:
                write = float(mt.group(2))
    if fetch is None:
        return None, None
    return (2.0 * fetch + (write or 0.0)) * 1024.0, "profiles/" + os.path.basename(files[-1]) + " (2 x FETCH_SIZE + WRITE_SIZE, KiB per dispatch)"


def cpu_baseline(A, W0, H0, k, iters, trace):
    from oracle import ref
    ref.lib()
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    r = ref.c_nnmf(A, k, W0, H0, None, None, [0, 0, 0], [0, 0, 0], iters, -1.0, 0, 0, False, INNER, INNER_TOL, METHOD, trace)
    dt = time.perf_counter() - t0
    return dict(value=iters / dt, unit="iterations/s", cores=cores, kind="port",
                sample=f"{iters} outer iterations of the same 20000x10000 k=50 MSE+SCD problem (same A, W0, H0, trace={trace}) "
                       f"with oracle/nnlm_ref.c (C/OpenMP restatement with the reference's cost structure, hand-written loops "
                       f"instead of BLAS), n.threads = all {cores} host threads",
                seconds=dt, final_mse=float(r["mse_error"][-1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--trace", type=int, default=2, help="error block every TRACE steps (R default 2 for MSE); 0 = never")
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-iters", type=int, default=2, help="outer iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--size", default=None, help="n,m,k override for quick experiments (reported in config)")
    args = ap.parse_args()

    n, m, k = (int(v) for v in args.size.split(",")) if args.size else (N_, M_, K_)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)

    import nnlm_amd
    from nnlm_amd import _lib

    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    A, W0, H0 = make_inputs(n, m, k)
    prec = _lib.PREC_F64 if args.precision == "f64" else _lib.PREC_F32
    h = nnlm_amd.Handle(local_rank, prec)
    if world > 1:
        ids = [_lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        h.comm_init(ids[0], rank, world)
    t0 = time.perf_counter()
    h.set_matrix(A)
    upload_s = time.perf_counter() - t0
    h.set_factors(k, W0, H0)

    def barrier():
        h.sync()
        if dist is not None:
            dist.barrier()
        h.sync()

    run_steps(h, args.warmup, args.trace, 0)
    barrier()
    t0 = time.perf_counter()
    mse = run_steps(h, args.steps, args.trace, args.warmup)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    final_mse = h.errors()[0] if mse is None else mse

    # replay the same K steps with per-kernel HIP events (on the library's own stream) for the roofline block
    h.profile_reset()
    h.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    run_steps(h, args.steps, args.trace, args.warmup + args.steps)
    barrier()
    prof_elapsed = time.perf_counter() - t0
    kern = {}
    for name in ("xprod_h", "xprod_w", "gram", "sweep_h", "sweep_w", "errors"):
        ms, cnt = h.profile_get(name)
        kern[name] = dict(ms_per_launch=(ms / cnt if cnt else None), launches=cnt, total_ms=ms)
    h.profile_enable(False)

    if rank != 0:
        h.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    s = 8 if args.precision == "f64" else 4
    # algorithmic HBM bytes of ONE cross-product launch (DESIGN.md "Roofline accounting"): A once, the fixed factor
    # once, the fp64 cross product once.  At N ranks each rank streams 1/N of A.
    bytes_h = (n * m * s + k * n * s) / world + k * m * 8
    bytes_w = (n * m * s + k * m * s) / world + k * n * 8
    dom = "xprod_w" if (kern["xprod_w"]["total_ms"] or 0) >= (kern["xprod_h"]["total_ms"] or 0) else "xprod_h"
    dom_bytes = bytes_w if dom == "xprod_w" else bytes_h
    dom_ms = kern[dom]["ms_per_launch"]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms else None
    x16 = s == 4 and os.environ.get("NNLM_XPROD", "") != "f32"   # split-fp16 cross products: one kernel for both half-steps
    if x16:  # the W half-step's launches mix xprod16_tn_kernel with the fused cross-product + error-block kernel: price the pure one
        dom, dom_bytes = "xprod_h", bytes_h
        dom_ms = kern[dom]["ms_per_launch"]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms else None
    kname = "xprod16_tn_kernel" if x16 else ("xprod_nt_kernel" if dom == "xprod_w" else "xprod_tn_kernel")
    traffic, traffic_src = pmc_traffic(kname) if world == 1 else (None, None)
    roofline = dict(bound="hbm", kernel=f"{dom} ({kname})", achieved=achieved,
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=(achieved / HBM_PEAK_GBS if achieved else None), traffic=traffic,
                    traffic_source=traffic_src,
                    bytes_per_launch=dom_bytes, ms_per_launch=dom_ms,
                    mfma_tflops=(2.0 * n * m * k / world / (dom_ms * 1e-3) / 1e12 if dom_ms else None))
    # the sweeps are a loop-carried recurrence (no HBM/MFMA roofline): reported as achieved fp64
    # arithmetic of the recurrence itself, inner*cols*k*(2k+8) flops per launch, against the fp64 vector peak
    sweep_info = {}
    for kname, cols in (("sweep_h", m), ("sweep_w", n)):
        ms_l = kern[kname]["ms_per_launch"]
        if ms_l:
            fl = INNER * (cols / world) * k * (2 * k + 8)
            sweep_info[kname] = dict(flops_per_launch=fl, tflops=fl / (ms_l * 1e-3) / 1e12, frac_of_fp64_peak=fl / (ms_l * 1e-3) / 1e12 / 78.6,
                                     note="latency bound: 2500 dependent coordinate steps per column")
    total_k = sum(v["total_ms"] for v in kern.values()) or 1.0
    shares = {kname: round(v["total_ms"] / total_k, 4) for kname, v in kern.items()}

    cpu = None
    if args.cpu_iters > 0 and world == 1:  # the CPU baseline is timed on rank 0 at N=1 only
        cpu = cpu_baseline(A, W0, H0, k, args.cpu_iters, args.trace if args.trace > 0 else 999999)

    out = {
        "metric": "nnmf iterations/sec + final MSE, dense A 20000x10000 k=50, 1/2/4/8 GPU",
        "value": args.steps / elapsed,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "final_mse": final_mse,
        "config": {"workload": f"nnmf(A, k={k}) MSE+SCD on {n}x{m} dense A (BASELINE.json configs[1])", "n": n, "m": m, "k": k,
                   "inner_max_iter": INNER, "inner_rel_tol": INNER_TOL, "trace": args.trace, "rel_tol": -1,
                   "arith": (("A fp32 (4 B/element); cross products: operands as split fp16 pairs (hi + lo*2^-11, 22 bits) on "
                              "v_mfma_f32_16x16x32_f16 with fp32 accumulation folded into fp64 every 256 elements"
                              if os.environ.get("NNLM_XPROD", "") != "f32" else
                              "A + cross-product GEMMs fp32 MFMA (fp64 flush every 256)") + "; Gram/mu/sweeps fp64" if s == 4
                             else "all fp64 (v_mfma_f64_16x16x4_f64)"),
                   "parallelism": (f"contraction sharded x{world} + 1 RCCL all-reduce, sweep sharded by columns + 1 all-gather, per half-step"
                                   if world > 1 else "1 GPU")},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "kernels": kern,
        "kernel_time_share": shares,
        "sweeps": sweep_info,
        "profiled_ms_per_step": 1e3 * prof_elapsed / args.steps,
        "upload_and_prep_s": upload_s,
    }
    print(json.dumps(out))
    h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
