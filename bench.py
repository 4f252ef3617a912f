#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on N GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # no WORLD_SIZE in the environment: spawns its N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one outer iteration of nnmf(): W half-step + H half-step (src/nnmf.cpp:114-133) plus, every `trace`
steps, the error block (src/nnmf.cpp:135-160).  Default workload = BASELINE.json configs[1]: nnmf(A, k=50), MSE loss,
sequential coordinate descent, dense A 20000 x 10000 = U(0,1) synthetic, explicit 0.01*U(0,1) init, R defaults
inner.max.iter=50, inner.rel.tol=1e-9, trace=2, rel.tol=-1 (fixed work).  A, W, H are resident in HBM when the timed
region starts.  N > 1: A replicated, every rank does the half-step of its 1/N of the columns, one RCCL all-gather per half-step
(strong scaling; `--form reduce`: contraction-sharded + all-reduce + all-gather; `--form both`, the default at N > 1: `value` is the
column form and `forms` carries step time and per-phase times of BOTH in the one line).
`--config 3` / `--config 5` time BASELINE.json configs[2] (KL + Lee) / configs[4] (10 % NA + L1/L2) the same way (second
bench lines for profiles/; the driver's line is the default config 2).  `--protocol core` = SURVEY section 8d's P-core (no
error block inside the timed iterations).

Prints ONE JSON line on rank 0 (see the task contract) with
  roofline            the kernel with the largest share of the step (HIP-event timed inside this process; the two half-steps' plain cross
                      products are launches of one kernel and count as one class): at config 2 the A-streaming cross product, HBM bound,
  roofline_secondary  the next class by time (config 2: the persistent SCD sweep, bound by dependent latency),
  other_configs       (N = 1, default config only) a few iterations each of the same problem in the strict fp64 mode (the .Call
                      boundary's default) and of BASELINE configs[2] (KL + Lee) and configs[4] (10 % NA + L1/L2) in the fp32-operand
                      mode: ms per step, dominant kernel class and its roofline fraction -- no extra CPU samples,
  step                whole-step fractions: SURVEY section 8d's bytes per iteration / time against the HBM peak, it/s against
                      its ceiling,
  cpu_baseline        oracle/nnlm_ref.c (OpenMP) on this box's host cores over the SAME iteration window (it starts from the
                      factors the GPU had after its warm-up), plus one full iteration on ONE thread (n.threads = 1 is R's default),
  call                (N = 1, default config) SURVEY 8d's call-level metric: n.iteration / wall time of ONE nnlm_c_nnmf() -- the .Call
                      entry: handle, code object, allocations, upload of A, 200 iterations, download -- from a cold process, both modes,
  mse_check           GPU and CPU mse after the same number of iterations from the same state.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # MI355X_MICROARCH.md: the copy rate measured on the part (what a streaming kernel can reach)
FP64_PEAK_TF = 78.6        # fp64 vector = matrix peak
FP16_MFMA_PEAK_TF = 2500.0 # v_mfma_f32_16x16x32_f16 / bf16, dense (MI355X_MICROARCH.md: ~2.5 PF)
# KL solvers: 3 plain + 1 transcendental fp32 instruction per element and coordinate = 14.9 cycles per 64 elements per SIMD
# measured in isolation (scripts/exp/valu_exp.hip, "KL body packed"): 1024 SIMDs x 64 / 14.9 x 2.4 GHz
KL_PEAK_GELEM = 1024 * 64 / 14.9 * 2.4
# strict fp64 KL: ~12 fp64 VALU instructions per element and coordinate (rcp + 2 Newton steps + residual correction, the sums, the
# refresh) at the fp64 vector rate of 16 lanes per cycle per SIMD
KL64_PEAK_GELEM = 1024 * 16 / 12.0 * 2.4
DTYPE_NAMES = {"f32": "f32 (A fp32; cross-product operands 22-bit split-fp16 pairs; Gram / sweeps fp64)", "f64": "f64"}
N_, M_, K_ = 20000, 10000, 50
INNER_TOL = float(os.environ.get("NNLM_BENCH_INNER_TOL", "1e-9"))  # (env: experiments only)
SEED = 20250928

CONFIGS = {
    2: dict(name="nnmf(A, k={k}) MSE+SCD on {n}x{m} dense A (BASELINE.json configs[1])", method=1, inner=50, trace=2, reg=[0.0, 0.0, 0.0], na=False),
    3: dict(name="nnmf(A, k={k}) KL loss + Lee multiplicative update on {n}x{m} dense A (BASELINE.json configs[2])", method=4, inner=1, trace=100,
            reg=[0.0, 0.0, 0.0], na=False),
    5: dict(name="nnmf(A, k={k}) MSE+SCD with 10% NA + L1/L2 reg on {n}x{m} A (BASELINE.json configs[4])", method=1, inner=50, trace=2,
            reg=[0.01, 0.0, 0.01], na=True),
}


def make_inputs(n, m, k, na):
    rng = np.random.default_rng(SEED)
    A = np.asfortranarray(rng.random((n, m)))  # (column-major, as R holds it: the binding then passes it without a transposing copy)
    W0 = 0.01 * rng.random((n, k))
    H0 = 0.01 * rng.random((k, m))
    if na:
        idx = np.random.default_rng(7).choice(n * m, n * m // 10, replace=False)  # (row-major linear indices, as in rounds 1-3)
        A[np.unravel_index(idx, (n, m))] = np.nan
    return A, W0, H0


def run_steps(h, cfg, steps, trace):
    """`steps` outer iterations of the reference loop (src/nnmf.cpp:109-161) on the resident problem: W half-step,
    H half-step, error block every `trace` iterations (+ the closing one), rel.tol = -1 so that no iteration is skipped.
    This is nnlm_run(), the same code path nnlm_c_nnmf() takes after its upload.  Returns the traces."""
    r = h.run(cfg["reg"], cfg["reg"], steps, -1.0, 0, False, cfg["inner"], INNER_TOL, cfg["method"], trace if trace > 0 else 999999)
    assert r["n_iteration"] == steps
    return r


def pmc_traffic(kernel, config=2, f64=False):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary of this configuration
    (profiles/rNN_<tag>_cfg<config>_pmc_hbm_summary.txt; round-1 files carry no cfg part and are configuration 2;
    separate --pmc FETCH_SIZE / WRITE_SIZE passes over this same command; KiB per dispatch).  gfx950 correction
    (MI355X_MICROARCH.md, HBM section): a wide streaming read is tallied at half its bytes, so reads = 2 x FETCH_SIZE.
    bench.py cannot attach rocprofv3 to itself: (None, None) when no summary holding the kernel is committed."""
    import glob, re
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_summary.txt"))
             if (f"_cfg{config}_" in os.path.basename(f) or ("_cfg" not in os.path.basename(f) and config == 2))
             and (("_f64_" in os.path.basename(f)) == f64)]  # (strict-mode summaries carry _f64_ in their name)
    files.sort(key=lambda f: [int(t) for t in re.findall(r"\d+", os.path.basename(f))])  # r01_v10 after r01_v9, r02 after r01
    for path in reversed(files):
        fetch = write = None
        for line in open(path):
            if kernel in line:
                mt = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+\s+mean=([0-9.eE+-]+)", line)
                if mt and mt.group(1) == "FETCH_SIZE":
                    fetch = float(mt.group(2))
                elif mt:
                    write = float(mt.group(2))
        if fetch is not None:
            return (2.0 * fetch + (write or 0.0)) * 1024.0, "profiles/" + os.path.basename(path) + " (2 x FETCH_SIZE + WRITE_SIZE, KiB per dispatch)"
    return None, None


def cpu_baseline(A, Ww, Hw, k, cfg, iters, trace):
    """oracle/nnlm_ref.c from the warmed factors (the window the GPU's timed region covers), all host threads; plus ONE whole
    outer iteration (A.t(), both half-steps, error block) on one thread -- R's default n.threads = 1 -- measured, not extrapolated."""
    from oracle import ref
    ref.lib()
    cores = os.cpu_count() or 1
    n, m = A.shape
    z = cfg["reg"]
    t0 = time.perf_counter()
    r = ref.c_nnmf(A, k, Ww, Hw, None, None, z, z, iters, -1.0, 0, 0, False, cfg["inner"], INNER_TOL, cfg["method"], trace)
    dt = time.perf_counter() - t0
    out = dict(value=iters / dt, unit="iterations/s", cores=cores, kind="port",
               sample=f"{iters} outer iterations of the same problem (same A; W, H as the GPU had them after its warm-up; trace={trace}) "
                      f"with oracle/nnlm_ref.c (C/OpenMP restatement with the reference's cost structure, hand-written loops "
                      f"instead of BLAS), n.threads = all {cores} host threads",
               seconds=dt, final_mse=float(r["mse_error"][-1]), average_epoch=float(np.sum(r["average_epoch"]) / max(iters, 1)))
    if os.environ.get("NNLM_BENCH_THREADS1", "1") == "1":
        t0 = time.perf_counter()
        r1 = ref.c_nnmf(A, k, Ww, Hw, None, None, z, z, 1, -1.0, 1, 0, False, cfg["inner"], INNER_TOL, cfg["method"], trace)
        t1 = time.perf_counter() - t0
        out["threads1"] = dict(value=1.0 / t1, unit="iterations/s", cores=1, seconds=t1, final_mse=float(r1["mse_error"][-1]),
                               sample="ONE whole outer iteration of the same problem from the same state (A.t(), W half-step, H half-step, error block) "
                                      "with oracle/nnlm_ref.c on one thread (R's default n.threads = 1): measured, not extrapolated")
    return out, r


def analyse(cfg, config_id, kern, n, m, k, s, world, sweep_forms=None):
    """Roofline blocks from the HIP-event scopes of one profiled run: (dominant class, secondary = the H half-step's cross product
    when it is not the dominant one, all classes, shares of kernel time).  s = bytes per stored element of A."""
    inner, method = cfg["inner"], cfg["method"]
    x16 = s == 4   # fp32-operand mode: split-fp16 cross products, one kernel for both half-steps
    # ---- algorithmic work per launch of each kernel class (DESIGN.md section 4 / SURVEY section 8d) ---------------------
    # cross products (HBM): A once, the fixed factor once, the fp64 cross product once; at N ranks each rank streams 1/N of A
    bytes_h = (n * m * s + k * n * s) / world + k * m * 8
    bytes_w = (n * m * s + k * m * s) / world + k * n * 8
    classes = {}
    if method < 3:
        for nm, by in (("xprod_h", bytes_h), ("xprod_w", bytes_w), ("xprod_w_err", bytes_w)):
            if kern[nm]["ms_per_launch"]:
                kname = ("xprod16_err_kernel" if nm == "xprod_w_err" else "xprod16_tn_kernel") if x16 else "xprod_tn_kernel"
                classes[nm] = dict(bound="hbm", kernel=f"{nm} ({kname})", work=by, peak=HBM_PEAK_GBS, unit="GB/s", scale=1e9, pmc=kname)
                if nm == "xprod_w_err":
                    classes[nm]["note"] = "W half-step cross product that also evaluates the error sums of the trace iteration (no separate pass over A); same algorithmic bytes"
        for nm, cols in (("sweep_h", m), ("sweep_w", n)):
            if kern[nm]["ms_per_launch"]:
                # the sweeps are a loop-carried recurrence (SURVEY 8d grants them no HBM roofline): priced as achieved fp64 arithmetic of
                # the recurrence, inner*cols*k*(2k+8) flops per launch, against the fp64 matrix peak -- the rank-4 gradient updates run on
                # v_mfma_f64_4x4x4 (16 of them per block of 4 coordinates and 16 columns), and 4x4x4 DGEMM issue is what bounds the kernel
                # (one wavefront per SIMD: dependent-issue latency of the chain; two: the matrix pipe, which a second wavefront cannot overlap)
                fl = inner * (cols / world) * k * (2 * k + 8)
                # (both modes since round 3: <.., STRICT = false / true>; round 4: between one and two 16-column wavefronts per SIMD -- 1024 SIMDs --
                #  the launch takes the persistent form sweep_scd_qw_kernel: 5 .. 7 column groups per CU shared by the wrap-around rule)
                # (which form the launch took is the library's decision -- device CU count, LDS limit: nnlm_get_info, recorded by the caller)
                form = (sweep_forms or {}).get(nm)
                sweep_kernel = {1: "sweep_scd_qw_kernel", 2: "sweep_scd_f_kernel", 3: "sweep_row_kernel"}.get(form, "sweep_scd_q_kernel")
                knm = ("na_gram_f16_kernel + colsolve_row_kernel" if s == 4 else "na_gram_lds_kernel + colsolve_strict_kernel") if cfg["na"] else sweep_kernel
                pk = FP64_PEAK_TF
                note = ("fp64 MFMA issue (v_mfma_f64_4x4x4, 16.6 cycles each, 16 per block-step of 16 columns) / dependent coordinate steps; "
                        "flops = inner*cols*k*(2k+8); a SIMD runs one 16-column wavefront at full speed, so the floor of a launch is "
                        "max(inner, ceil(groups per CU * inner / 4)) sweeps of ~2.16 us")
                if form == 3 and not cfg["na"]:
                    # row form (k_sweep_r.h): four columns per wavefront, five fp32 vector instructions per step of four columns, DPP row broadcast;
                    # taken while the launch is one round of its wavefronts (<= 32 columns per CU) -- mid-size problems, multi-GPU shards
                    pk = 157.3
                    note = ("fp32 chain, row form (k_sweep_r.h): 5 vector instructions per coordinate step of 4 columns, 12.8 ns per step for a lone wavefront, "
                            "19.5 ns each for two per SIMD (scripts/exp/lane_exp.hip); flops = inner*cols*k*(2k+8) against the fp32 vector peak for scale")
                elif form == 2 and not cfg["na"]:
                    # round 6, fp32-operand mode: fp32 chain state, rank-4 updates as three-piece bf16 products on v_mfma_f32_16x16x32_bf16.
                    # Bound: instruction issue of ONE wavefront per SIMD (6.5 cycles per instruction, ~34 per block of 4 coordinates and
                    # 16 columns: scripts/exp/issue_exp.hip, profiles/r06_issue_exp.log) -- reported against the fp32 vector/matrix peak for scale
                    pk = 157.3
                    note = ("fp32 chain (k_sweep_f.h): ~34 instructions per block of 4 coordinates x 16 columns at one instruction per 6.5 cycles "
                            "from a lone wavefront (240 cycles per block at 2.4 GHz in isolation, ~300 at the ~1.9 GHz the clocks sit at behind the "
                            "A-streaming kernels); flops = inner*cols*k*(2k+8) against the 157.3 TF fp32 peak, for scale only")
                if cfg["na"]:
                    # + the per-column Grams over the complement rows, 2 k^2 flops per missing entry.  F32 mode: they run on the fp16 matrix
                    # cores as three split-fp16 products (a third of the dense fp16 peak per algorithmic flop); strict mode: fp64 MFMA.
                    # The scope holds both kernels, so the peak is the blend that gives frac = (t_gram_floor + t_solver_floor) / t_measured.
                    gfl = 2.0 * k * k * (n * m // 10)
                    gpk = FP16_MFMA_PEAK_TF / 3.0 if s == 4 else FP64_PEAK_TF
                    floor_s = gfl / (gpk * 1e12) + fl / (FP64_PEAK_TF * 1e12)
                    fl += gfl
                    pk = fl / floor_s / 1e12
                    note += (" (fp64 vector peak) + 2k^2 per missing entry for the per-column Grams ("
                             + ("3 split-fp16 products per flop on v_mfma_f32_16x16x32_f16: a third of the 2.5 PF dense fp16 peak" if s == 4 else "fp64 MFMA peak")
                             + "); peak = total flops / (sum of the two floors)")
                # dense sweep: a loop-carried recurrence, bound by the dependent-issue latency of its 4 x ceil(k/4) x inner stages per
                # column (SURVEY 8d grants it no roofline); the fraction of the fp64 matrix peak is reported for scale, not as its bound.
                # NA flow: per-column Grams on the matrix cores + the solver: "mfma".
                classes[nm] = dict(bound=("mfma" if cfg["na"] else "latency"), kernel=f"{nm} ({knm})", work=fl, peak=pk, unit="TFLOP/s", scale=1e12,
                                   pmc=(None if cfg["na"] else sweep_kernel), note=note)
    else:
        for nm in ("sweep_h", "sweep_w"):
            if kern[nm]["ms_per_launch"]:
                el = float(n) * m * k * inner  # element-steps: every entry of A meets every coordinate once per sweep
                if s == 4:
                    classes[nm] = dict(bound="valu", kernel=f"{nm} (kl_tile_kernel)", work=el, peak=KL_PEAK_GELEM, unit="Gelem/s", scale=1e9, pmc=None,
                                       note="fp32 VALU: 3 plain + 1 v_rcp_f32 per element and coordinate; peak = 14.9 cycles per 64 elements per SIMD "
                                            "(scripts/exp/valu_exp.hip)")
                else:
                    classes[nm] = dict(bound="valu", kernel=f"{nm} (kl_reg64_kernel)", work=el, peak=KL64_PEAK_GELEM, unit="Gelem/s", scale=1e9, pmc=None,
                                       note="fp64 VALU: ~12 fp64 instructions (correctly rounded quotient) per element and coordinate at 16 lanes per cycle per SIMD")
    if kern["errors"]["ms_per_launch"] and "errors" not in classes:
        classes["errors"] = dict(bound="hbm", kernel="errors (errors_f32_kernel / errors64_kernel: a separate pass over A)", work=n * m * s / world,
                                 peak=HBM_PEAK_GBS, unit="GB/s", scale=1e9, pmc=None)

    def block(nm):
        c = classes[nm]
        ms_l = kern[nm]["ms_per_launch"]
        ach = c["work"] / (ms_l * 1e-3) / c["scale"]
        traffic, src = (pmc_traffic(c["pmc"], config_id, s == 8) if (c["pmc"] and world == 1) else (None, None))
        b = dict(bound=c["bound"], kernel=c["kernel"], achieved=ach, peak=c["peak"], unit=c["unit"], frac=ach / c["peak"], traffic=traffic,
                 traffic_source=src, work_per_launch=c["work"], ms_per_launch=ms_l, share_of_kernel_time=None)
        if c["bound"] == "hbm":  # next to the 8 TB/s pin rate: the copy rate the microarchitecture guide measured on this part
            b["frac_of_achievable"] = ach / HBM_ACHIEVABLE_GBS
            b["achievable"] = HBM_ACHIEVABLE_GBS
        if "note" in c:
            b["note"] = c["note"]
        return b

    total_k = sum(v["total_ms"] for v in kern.values()) or 1.0
    shares = {kname: round(v["total_ms"] / total_k, 4) for kname, v in kern.items()}
    # The two half-steps' plain cross products are launches of ONE kernel (xprod16_tn_kernel on A16 / A16T, xprod_tn_kernel<double> on A / AT):
    # the row rocprofv3's per-kernel summary ranks first.  For the ranking they are one class, "xprod" -- algorithmic bytes and time per
    # launch averaged over its launches --, so that `roofline` is the dominant KERNEL (profiles/rNN_cfg2_kernel_stats.csv, first row), not
    # the largest per-half-step scope.
    kern = dict(kern)
    merged = set()
    if "xprod_h" in classes and "xprod_w" in classes:
        kh, kw = kern["xprod_h"], kern["xprod_w"]
        nl = kh["launches"] + kw["launches"]
        kern["xprod"] = dict(ms_per_launch=(kh["total_ms"] + kw["total_ms"]) / nl, launches=nl, total_ms=kh["total_ms"] + kw["total_ms"])
        ch = classes["xprod_h"]
        classes["xprod"] = dict(ch, kernel=f"xprod_h + xprod_w ({ch['pmc']}: both half-steps' plain launches)",
                                work=(classes["xprod_h"]["work"] * kh["launches"] + classes["xprod_w"]["work"] * kw["launches"]) / nl)
        shares["xprod"] = round(kern["xprod"]["total_ms"] / total_k, 4)
        merged = {"xprod_h", "xprod_w"}
    ranked = sorted((nm for nm in classes if nm != "errors" and nm not in merged), key=lambda nm: -kern[nm]["total_ms"])
    roofline = block(ranked[0]) if ranked else None
    if roofline:
        roofline["share_of_kernel_time"] = shares[ranked[0]]
    secondary = None
    if len(ranked) > 1:  # the next class by time (config 2: the persistent SCD sweep -- a loop-carried recurrence, no HBM / MFMA roofline: "latency")
        secondary = block(ranked[1])
        secondary["share_of_kernel_time"] = shares[ranked[1]]
    all_blocks = {nm: block(nm) for nm in classes}
    return roofline, secondary, all_blocks, shares


SCOPES = ("xprod_h", "xprod_w", "xprod_w_err", "gram", "sweep_h", "sweep_w", "errors", "err_reduce", "allgather", "allreduce", "unpack")
# (errors: a separate pass over A -- errors_f32_kernel / errors64_kernel; err_reduce: the 5 us reduction of the partial sums the fused cross
#  product xprod16_err_kernel left behind, no HBM work of its own -- not priced against a roofline;
#  allgather / allreduce: the RCCL collective of a sharded half-step between two HIP events on the stream it is enqueued on -- includes
#  the wait for the slowest rank; unpack: shard_unpack_kernel + the sum of the ranks' Gram partial sums)


def profile_scopes(h):
    kern = {}
    for name in SCOPES:
        ms, cnt = h.profile_get(name)
        kern[name] = dict(ms_per_launch=(ms / cnt if cnt else None), launches=cnt, total_ms=ms)
    return kern


def sweep_forms_of(h):
    """{"sweep_w": 0 | 1 | 2 | -1, "sweep_h": ...}: form of the handle's last SCD sweep launches (0 plain fp64 chain, 1 persistent fp64 chain,
    2 fp32 chain -- the fp32-operand mode since round 6; nnlm_get_info)."""
    return {"sweep_w": int(h.get_info("sweep_form_w")), "sweep_h": int(h.get_info("sweep_form_h"))}


def call_probe(precision, n, m, k, max_iter):
    """(child process of `call_metric`) ONE nnlm_c_nnmf() from a cold process state, then a second one: wall seconds of each."""
    from nnlm_amd import _lib
    cfg = CONFIGS[2]
    A, W0, H0 = make_inputs(n, m, k, False)
    z = cfg["reg"]
    res = []
    for _ in range(2):
        t0 = time.perf_counter()
        r = _lib.c_nnmf(A, k, W0, H0, None, None, z, z, max_iter, -1.0, 1, 0, False, cfg["inner"], INNER_TOL, cfg["method"], cfg["trace"])
        res.append(dict(wall_s=time.perf_counter() - t0, n_iteration=r["n_iteration"], final_mse=float(r["mse_error"][-1])))
    print(json.dumps(dict(precision=precision, max_iter=max_iter, cold=res[0], warm=res[1])), flush=True)


def call_metric(n, m, k, local_rank, max_iter=200):
    """SURVEY section 8d's call-level metric (R/nnmf.R:176-183, what $run.time covers): n.iteration / wall seconds of ONE
    nnlm_c_nnmf() -- create, code-object load, allocations, upload of A, `max_iter` iterations with R's trace = 2, download --
    in a FRESH process per arithmetic mode (cold: nothing of the library has run in it), and of a second call in the same process."""
    import subprocess
    out = {}
    for prec in ("f32", "f64"):
        env = dict(os.environ, NNLM_PRECISION=prec, NNLM_DEVICE=str(local_rank))
        for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(v, None)
        cmd = [sys.executable, os.path.abspath(__file__), "--call-probe", prec, "--size", f"{n},{m},{k}", "--call-iters", str(max_iter)]
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not line:
                out[prec] = dict(error=f"rc {p.returncode}: {p.stderr.strip()[-300:]}")
                continue
            d = json.loads(line[-1])
            for key in ("cold", "warm"):
                d[key]["iterations_per_s"] = d[key]["n_iteration"] / d[key]["wall_s"]
            out[prec] = dict(dtype=DTYPE_NAMES[prec], max_iter=max_iter, cold=d["cold"], warm=d["warm"])
        except Exception as e:  # (never lose the headline over a secondary measurement)
            out[prec] = dict(error=f"{type(e).__name__}: {e}")
    out["note"] = ("wall time of one nnlm_c_nnmf() call (the .Call entry: handle + code object + allocations + upload of the fp64 matrix + "
                   f"{max_iter} iterations at trace = 2, rel.tol = -1 + download), python ctypes wrapper included; cold = first call of a fresh "
                   "process, warm = second call in the same process; f64 is the .Call default (NNLM_PRECISION unset)")
    return out


def self_launch(n_ranks):
    """`bench.py --gpus N` without a launcher: spawn the N ranks (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
    as torch.distributed.run would), relay rank 0's stdout -- the one JSON line -- and return the first non-zero exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (the host driver only supports dmabuf IPC: RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(subprocess.PIPE if r == 0 else subprocess.DEVNULL)))
    rc, out0 = 0, b""
    live = list(range(n_ranks))
    while live:  # a rank that dies takes the others with it (they would wait in a collective for ever)
        for r in list(live):
            try:
                if r == 0:
                    o, _ = procs[0].communicate(timeout=0.5)
                    out0 += o or b""
                else:
                    procs[r].wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            live.remove(r)
            if procs[r].returncode != 0 and rc == 0:
                rc = procs[r].returncode
                print(f"bench.py: rank {r} of {n_ranks} exited with {rc}; stopping the others", file=sys.stderr)
                for q in live:
                    procs[q].terminate()
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return rc


def step_block(cfg, n, m, k, s, trace, ms_step):
    """The whole step against SURVEY section 8d's per-iteration figures.  s = bytes per stored element of A (4: fp32-operand mode, 8: strict)."""
    inner, method = cfg["inner"], cfg["method"]
    b_it = 2.0 * n * m * s + 4.0 * k * (n + m) * s + (n * m * s / trace if trace > 0 else 0.0)
    hbm_ms = b_it / (HBM_PEAK_GBS * 1e9) * 1e3
    if method < 3 and not cfg["na"]:   # configs 2 / 4: two A-streaming skinny GEMMs per iteration
        f_it = 4.0 * n * m * k + 4.0 * k * k * (n + m)
        ceil_ms = max(hbm_ms, f_it / ((157.3e12 if s == 4 else 78.6e12)) * 1e3)
        note = "SURVEY 8d: B_it = 2nm s + 4k(n+m)s (+ nm s per trace iteration); ceiling = max(B_it / 8 TB/s, F_it / MFMA peak)"
    elif method >= 3:                  # config 3: KL solvers, sequential in k -- VALU / transcendental rate, not HBM
        f_it = 2.0 * 7.0 * n * m * k * inner
        ceil_ms = max(hbm_ms, f_it / ((157.3e12 if s == 4 else 78.6e12)) * 1e3)
        note = ("SURVEY 8d: VALU work 2 x 7 n m k inner flop-equivalents per iteration (incl. 2 n m k inner divides) against the "
                + ("157.3 TF fp32" if s == 4 else "78.6 TF fp64") + " vector peak; bytes as config 2")
    else:                              # config 5: per-column Grams over the complement rows on the matrix cores
        nmiss = n * m // 10
        f_it = 2.0 * 2.0 * k * k * min(nmiss, n * m - nmiss)
        pk = (FP16_MFMA_PEAK_TF / 3.0 if s == 4 else FP64_PEAK_TF) * 1e12
        ceil_ms = max(hbm_ms, f_it / pk * 1e3)
        note = ("SURVEY 8d: Gram flops 2 k^2 min(nnz, nm - nnz) per half-step on "
                + ("the fp16 matrix cores as three split products (2.5 PF / 3)" if s == 4 else "the fp64 matrix cores (78.6 TF)") + "; bytes as config 2")
    return dict(bytes_per_iteration=b_it, flops_per_iteration=f_it, hbm_frac=b_it / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, ceiling_ms=ceil_ms,
                ceiling_it_per_s=1e3 / ceil_ms, frac_of_ceiling=ceil_ms / ms_step, note=note)


def other_config(config_id, precision, n, m, k, local_rank, steps, warmup):
    """A short run of another configuration on the same device (N = 1): ms per step, dominant kernel class, its fraction."""
    import nnlm_amd
    from nnlm_amd import _lib
    cfg = dict(CONFIGS[config_id])
    A, W0, H0 = make_inputs(n, m, k, cfg["na"])
    with nnlm_amd.Handle(local_rank, _lib.PREC_F64 if precision == "f64" else _lib.PREC_F32) as h:
        h.set_matrix(A)
        del A
        h.set_factors(k, W0, H0)
        trace = cfg["trace"]
        run_steps(h, cfg, warmup, trace)
        h.sync()
        t0 = time.perf_counter()
        r = run_steps(h, cfg, steps, trace)
        h.sync()
        dt = time.perf_counter() - t0
        h.profile_reset()
        h.profile_enable(True)
        run_steps(h, cfg, steps, trace)
        h.sync()
        kern = profile_scopes(h)
        h.profile_enable(False)
        forms = sweep_forms_of(h)
    roof, _, blocks, shares = analyse(cfg, config_id, kern, n, m, k, 8 if precision == "f64" else 4, 1, forms)
    return dict(workload=cfg["name"].format(n=n, m=m, k=k), dtype=DTYPE_NAMES[precision], steps=steps, warmup=warmup, trace=trace,
                ms_per_step=1e3 * dt / steps, iterations_per_s=steps / dt, final_mse=float(r["mse_error"][-1]),
                step=step_block(cfg, n, m, k, 8 if precision == "f64" else 4, trace, 1e3 * dt / steps),
                dominant=dict(kernel=roof["kernel"], bound=roof["bound"], frac=roof["frac"], ms_per_launch=roof["ms_per_launch"],
                              share_of_kernel_time=roof["share_of_kernel_time"]) if roof else None,
                kernels_ms_per_launch={kk: v["ms_per_launch"] for kk, v in kern.items() if v["ms_per_launch"]},
                roofline_frac={kk: round(v["frac"], 4) for kk, v in blocks.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="2 (default, the headline), 3 (KL+Lee), 5 (NA + reg)")
    ap.add_argument("--trace", type=int, default=None, help="error block every TRACE steps (R default: 2 for MSE, 100 for MKL); 0 = never")
    ap.add_argument("--protocol", default="default", choices=["default", "core"], help="core: no error block inside the timed iterations (SURVEY 8d)")
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-iters", type=int, default=2, help="outer iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions (the first one is `value`; all are listed)")
    ap.add_argument("--size", default=None, help="n,m,k override for quick experiments (reported in config)")
    ap.add_argument("--others", type=int, default=1, help="1: after the headline (N = 1, default config) also time strict-fp64 config 2 and fp32 configs 3 and 5 "
                                                          "for a few iterations each -> other_configs; 0: skip")
    ap.add_argument("--form", default=os.environ.get("NNLM_SHARD_DENSE", "both"), choices=["cols", "reduce", "both"],
                    help="N > 1, dense square loss: cols = column-sharded half-steps + one all-gather each (`value` of `both`), reduce = contraction-sharded "
                         "+ one all-reduce of [G | C] + all-gather, both = time the two forms in one invocation -> forms")
    ap.add_argument("--call", type=int, default=1, help="1: (N = 1, default config) also time one cold nnlm_c_nnmf() call per mode in a child process -> call")
    ap.add_argument("--call-iters", type=int, default=200, help="max.iter of the call-level metric (SURVEY 8d: 200)")
    ap.add_argument("--call-probe", default=None, choices=["f32", "f64"], help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.call_probe:  # child of call_metric(): nothing but the one-shot call
        n_, m_, k_ = (int(v) for v in args.size.split(",")) if args.size else (N_, M_, K_)
        call_probe(args.call_probe, n_, m_, k_, args.call_iters)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # plain `python bench.py --gpus N`: be the launcher
        sys.exit(self_launch(args.gpus))

    # stdout carries exactly ONE line, the JSON: whatever libraries print on the way (RCCL's version banner, gloo's connection notes --
    # C stdio and Python alike) goes to stderr until the line is ready
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    cfg = dict(CONFIGS[args.config])
    n, m, k = (int(v) for v in args.size.split(",")) if args.size else (N_, M_, K_)
    trace = cfg["trace"] if args.trace is None else args.trace
    if args.protocol == "core":
        trace = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s), as the launcher set up", file=sys.stderr)

    import nnlm_amd
    from nnlm_amd import _lib

    # NNLM_BENCH_FORCE_COMM=1: build the communicator and take the sharded code path with ONE rank (tests the N > 1 plumbing --
    # id broadcast, LOCAL_RANK -> device, barriers, max-over-ranks -- on a one-GPU box)
    force_comm = os.environ.get("NNLM_BENCH_FORCE_COMM", "") == "1"
    dist = None
    if world > 1 or force_comm:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    A, W0, H0 = make_inputs(n, m, k, cfg["na"])
    prec = _lib.PREC_F64 if args.precision == "f64" else _lib.PREC_F32
    h = nnlm_amd.Handle(local_rank, prec)
    if dist is not None:
        ids = [_lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        h.comm_init(ids[0], rank, world, form=("reduce" if args.form == "reduce" else "cols"))
    t0 = time.perf_counter()
    h.set_matrix(A)
    upload_s = time.perf_counter() - t0
    h.set_factors(k, W0, H0)

    def barrier():
        h.sync()
        if dist is not None:
            dist.barrier()
        h.sync()

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    if os.environ.get("NNLM_BENCH_PRIME", "") == "1":  # (experiment: first launch of the separate error kernels before the timed region)
        h.errors()
    run_steps(h, cfg, args.warmup, trace)
    times, mses = [], []
    for _ in range(max(args.repeats, 1)):
        barrier()
        t0 = time.perf_counter()
        r = run_steps(h, cfg, args.steps, trace)
        barrier()
        times.append(max_over_ranks(time.perf_counter() - t0))
        mses.append(float(r["mse_error"][-1]))
    elapsed, final_mse = times[0], mses[0]

    # replay the same K steps with per-kernel HIP events (on the library's own stream) for the roofline block
    h.profile_reset()
    h.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    run_steps(h, cfg, args.steps, trace)
    barrier()
    prof_elapsed = time.perf_counter() - t0
    kern = profile_scopes(h)
    h.profile_enable(False)
    # per-phase time of a step, MAX over ranks (one SCALE line then shows where the step goes: what shards, what every rank repeats,
    # what the collectives cost)
    phase_tot = [kern[nm]["total_ms"] for nm in SCOPES]
    if dist is not None:
        import torch
        tt = torch.tensor(phase_tot, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        phase_tot = [float(v) for v in tt]
    phases_ms = {nm: v / args.steps for nm, v in zip(SCOPES, phase_tot) if v > 0}
    main_forms = sweep_forms_of(h)

    # --form both (N > 1, dense square loss): the same steps once more in the all-reduce form north_star words -- one timed region and
    # one profiled replay -- so that ONE line prices the [G | C] all-reduce against the all-gather-only default
    forms = None
    if dist is not None and args.form == "both" and cfg["method"] < 3 and not cfg["na"] and k <= 64:
        forms = {"cols": dict(ms_per_step=1e3 * times[0] / args.steps, phases_ms=dict(phases_ms))}
        h.comm_set_form("reduce")
        run_steps(h, cfg, args.warmup, trace)
        barrier()
        t0 = time.perf_counter()
        rr = run_steps(h, cfg, args.steps, trace)
        barrier()
        t_red = max_over_ranks(time.perf_counter() - t0)
        h.profile_reset()
        h.profile_enable(True)
        barrier()
        run_steps(h, cfg, args.steps, trace)
        barrier()
        kern_r = profile_scopes(h)
        h.profile_enable(False)
        tot_r = [kern_r[nm]["total_ms"] for nm in SCOPES]
        import torch
        tt = torch.tensor(tot_r, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        forms["reduce"] = dict(ms_per_step=1e3 * t_red / args.steps, final_mse=float(rr["mse_error"][-1]),
                               phases_ms={nm: float(v) / args.steps for nm, v in zip(SCOPES, tt) if float(v) > 0})
        forms["note"] = ("cols (`value`): a rank forms the cross product of ITS columns over the whole contraction, sweeps them, ONE all-gather per "
                         "half-step; reduce (north_star's wording): contraction-sharded [Gram | cross product], ONE all-reduce, column-sharded "
                         "sweep, ONE all-gather; same steps, the reduce region follows the cols regions from the factors they left")
        h.comm_set_form("cols")

    # GPU mse after `cpu_iters` iterations from the warmed state (what the CPU sample reproduces)
    gpu_check = None
    Ww = Hw = None
    if args.cpu_iters > 0 and world == 1 and not force_comm:
        # the state the timed region started from, reproduced (the loop is deterministic): the CPU sample starts there too.
        # (Taken here, not between warm-up and timing: a host round trip there lets the clocks drop before the timed region.)
        h.set_factors(k, W0, H0)
        run_steps(h, cfg, args.warmup, trace)
        Ww, Hw = h.get_factors()
        gpu_check = run_steps(h, cfg, args.cpu_iters, trace)

    if rank != 0:
        h.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    s = 8 if args.precision == "f64" else 4
    inner, method = cfg["inner"], cfg["method"]
    roofline, secondary, all_blocks, shares = analyse(cfg, args.config, kern, n, m, k, s, world, main_forms)

    ms_step = 1e3 * elapsed / args.steps
    step = step_block(cfg, n, m, k, s, trace, ms_step)

    cpu = mse_check = None
    if args.cpu_iters > 0 and world == 1 and not force_comm:  # the CPU baseline is timed on rank 0 at N=1 only
        cpu, rc = cpu_baseline(A, Ww, Hw, k, cfg, args.cpu_iters, trace if trace > 0 else 999999)
        g, c_ = float(gpu_check["mse_error"][-1]), float(rc["mse_error"][-1])
        mse_check = dict(iterations_from_warm_state=args.cpu_iters, gpu=g, cpu=c_, rel_diff=abs(g - c_) / c_,
                         gpu_average_epoch=float(np.sum(gpu_check["average_epoch"]) / args.cpu_iters), cpu_average_epoch=cpu["average_epoch"])

    ms_all = [1e3 * t / args.steps for t in times]
    # the rest of the picture in the same line (task 3): the .Call boundary's default mode and the other single-GPU configs
    others = None
    if args.others and world == 1 and not force_comm and args.config == 2 and args.precision == "f32" and args.protocol == "default":
        h.close()
        others = {}
        for key, cid, pr, st, wu in (("config2_strict_f64", 2, "f64", 16, 3), ("config3_kl_lee_f32", 3, "f32", 8, 2), ("config5_na_reg_f32", 5, "f32", 12, 2)):
            try:
                others[key] = other_config(cid, pr, n, m, k, local_rank, st, wu)
            except Exception as e:  # (never lose the headline over a secondary measurement)
                others[key] = dict(error=f"{type(e).__name__}: {e}")
    call = None
    if args.call and world == 1 and not force_comm and args.config == 2 and args.precision == "f32" and args.protocol == "default":
        del A  # (the child processes generate their own copy: 1.6 GB each way)
        call = call_metric(n, m, k, local_rank, args.call_iters)
    out = {
        "metric": "nnmf iterations/sec + final MSE, dense A 20000x10000 k=50, 1/2/4/8 GPU",
        "value": args.steps / elapsed,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": DTYPE_NAMES[args.precision],
        "data": "synthetic",
        "final_mse": final_mse,
        "config": {"workload": cfg["name"].format(n=n, m=m, k=k), "n": n, "m": m, "k": k, "method": method,
                   "inner_max_iter": inner, "inner_rel_tol": INNER_TOL, "trace": trace, "protocol": args.protocol, "rel_tol": -1,
                   "reg": cfg["reg"],
                   "arith": ("A fp32 (4 B/element); cross products: operands as split fp16 pairs (hi + lo*2^-11, 22 bits) on "
                             "v_mfma_f32_16x16x32_f16 with fp32 accumulation folded into fp64 every 256 elements; Gram/mu/sweeps fp64; "
                             "KL solvers fp32 state" if s == 4 else "all fp64 (v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4)"),
                   "parallelism": ((f"contraction sharded x{world} + 1 RCCL all-reduce, sweep sharded by columns + 1 all-gather, per half-step"
                                    if args.form == "reduce" and method < 3 and not cfg["na"] else
                                    f"columns sharded x{world} (cross product, Gram, sweep of a rank's columns) + 1 RCCL all-gather per half-step")
                                   if (world > 1 or force_comm) else "1 GPU")},
        "repeats": {"ms_per_step": ms_all, "min": min(ms_all), "median": float(np.median(ms_all)), "max": max(ms_all),
                    "final_mse": mses, "note": "consecutive timed regions of `steps` iterations each; `value` is the first"},
        "roofline": roofline,
        "roofline_secondary": secondary,
        "roofline_all": all_blocks,
        "other_configs": others,
        "forms": forms,
        "call": call,
        "step": step,
        "cpu_baseline": cpu,
        "mse_check": mse_check,
        "phases_ms": dict(phases_ms, note="HIP-event time per step of each phase in the profiled replay, max over ranks; 'gram' = Gram fold + split copy; "
                                           "'err_reduce' = reduction of the fused error sums; N > 1: 'allgather' / 'allreduce' = the RCCL call between two events on its stream "
                                           "(includes waiting for the slowest rank), 'unpack' = scatter of the gathered slabs (+ split copy + Gram sum)"),
        "kernels": kern,
        "kernel_time_share": shares,
        "profiled_ms_per_step": 1e3 * prof_elapsed / args.steps,
        "upload_and_prep_s": upload_s,
    }
    h.close()   # (communicator and process group go first: anything RCCL still prints must not follow the JSON line)
    if dist is not None:
        dist.destroy_process_group()
    try:  # RCCL prints its version banner through C stdio: flush it (to stderr, see above) before stdout comes back
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
