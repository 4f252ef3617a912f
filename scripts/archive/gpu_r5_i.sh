#!/bin/bash
# round 5, call I: KL low-memory fallback test, wrap policy/timing tests, larger-shape bench lines (generalised sweep), reduce-form unpack
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_edges.py tests/test_gpu_wrap.py -x -q -m gpu -s -k "streaming or policy or scales" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r05/i_tests.log
python bench.py --steps 20 --warmup 5 --size 40000,10000,50 --cpu-iters 0 --others 0 --call 0 > gpurun_out/r05/i_bench_40000.json 2> gpurun_out/r05/i_bench_40000.err
python bench.py --steps 20 --warmup 5 --size 29000,10000,50 --cpu-iters 0 --others 0 --call 0 > gpurun_out/r05/i_bench_29000.json 2>> gpurun_out/r05/i_bench_40000.err
NNLM_BENCH_FORCE_COMM=1 python bench.py --steps 20 --warmup 5 --cpu-iters 0 > gpurun_out/r05/i_bench_forced.json 2> gpurun_out/r05/i_bench_forced.err
cat gpurun_out/r05/i_tests.log
python - <<'PY'
import json
for f in ("i_bench_40000", "i_bench_29000", "i_bench_forced"):
    try:
        d = json.load(open(f"gpurun_out/r05/{f}.json"))
        print(f, round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["phases_ms"].items() if k != "note"}, d["roofline"]["kernel"])
        if d.get("forms"):
            print("   forms:", {k: (round(v["ms_per_step"], 4), {kk: round(vv, 4) for kk, vv in v["phases_ms"].items()}) for k, v in d["forms"].items() if k != "note"})
    except Exception as e:
        print(f, "ERR", e)
PY
