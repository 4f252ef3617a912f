#!/bin/bash
# round 5, call E: KL kernels after the step-loop rearrangement (tests + config 3 line), strict mode with the error block beside the sweep / beside
# the cross product, cold-call breakdown
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
python -m pytest tests -x -q -m gpu -k "kl or KL or config3 or method" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r05/e_tests_kl.log
python bench.py --config 3 --steps 8 --warmup 2 --cpu-iters 0 > gpurun_out/r05/e_bench_cfg3.json 2> gpurun_out/r05/e_bench_cfg3.err
python bench.py --precision f64 --steps 20 --warmup 5 --cpu-iters 0 --others 0 --call 0 > gpurun_out/r05/e_bench_f64_late.json 2>/dev/null
NNLM_EXP_ERR_EARLY=1 python bench.py --precision f64 --steps 20 --warmup 5 --cpu-iters 0 --others 0 --call 0 > gpurun_out/r05/e_bench_f64_early.json 2>/dev/null
python scripts/gpu_call_breakdown.py f32 > gpurun_out/r05/e_call_f32.json 2>/dev/null
python scripts/gpu_call_breakdown.py f64 > gpurun_out/r05/e_call_f64.json 2>/dev/null
cat gpurun_out/r05/e_tests_kl.log
python - <<'PY'
import json
for f in ("e_bench_cfg3", "e_bench_f64_late", "e_bench_f64_early"):
    try:
        d = json.load(open(f"gpurun_out/r05/{f}.json"))
        print(f, round(d["ms_per_step"], 4), d["repeats"]["ms_per_step"], {k: round(v, 4) for k, v in d["phases_ms"].items() if k != "note"})
    except Exception as e:
        print(f, "ERR", e)
for f in ("e_call_f32", "e_call_f64"):
    print(open(f"gpurun_out/r05/{f}.json").read().strip())
PY
