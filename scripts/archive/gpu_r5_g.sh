#!/bin/bash
# round 5, call G: wrap tests (policy + timing assertion), NA path after the five-instruction solver step (tests + config 5 line)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_wrap.py -x -q -m gpu -s -k "policy or scales" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r05/g_tests_wrap.log
python -m pytest tests -x -q -m gpu -k "missing or na_ or _na or NA or config5 or nsclc or fuzz" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r05/g_tests_na.log
python bench.py --config 5 --steps 12 --warmup 2 --cpu-iters 0 > gpurun_out/r05/g_bench_cfg5.json 2> gpurun_out/r05/g_bench_cfg5.err
cat gpurun_out/r05/g_tests_wrap.log gpurun_out/r05/g_tests_na.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/g_bench_cfg5.json"))
print(d["ms_per_step"], d["repeats"]["ms_per_step"], {k: round(v, 4) for k, v in d["phases_ms"].items() if k != "note"})
print({k: (round(v["ms_per_launch"], 4) if v["ms_per_launch"] else None) for k, v in d["kernels"].items()})
PY
