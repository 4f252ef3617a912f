#!/bin/bash
# round 5, call R: strict mode with a speculative H half-step behind the speculative W half-step (error block beside both) -- tests, bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | grep -v "$F" | tail -15 > gpurun_out/r05/r_tests.log
B="python bench.py --cpu-iters 0 --others 0 --call 0 --precision f64"
$B > gpurun_out/r05/r_bench_f64.json 2> gpurun_out/r05/r_bench_f64.err
$B --steps 20 --warmup 5 > gpurun_out/r05/r_bench_f64_steps20.json 2> gpurun_out/r05/r_bench_f64_steps20.err
$B --config 5 > gpurun_out/r05/r_bench_f64_cfg5.json 2> gpurun_out/r05/r_bench_f64_cfg5.err
cat gpurun_out/r05/r_tests.log
python - <<'PY'
import json
for f in ("r_bench_f64", "r_bench_f64_steps20", "r_bench_f64_cfg5"):
    try:
        d = json.load(open(f"gpurun_out/r05/{f}.json"))
        print(f, round(d["value"], 1), round(d["ms_per_step"], 4), [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["phases_ms"].items() if k != "note"}, "mse", d["final_mse"])
    except Exception as e:
        print(f, "ERR", e)
PY
