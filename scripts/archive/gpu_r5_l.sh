#!/bin/bash
# round 5, call L: the wavefront-specialised fused cross product / error block (xprod16_err_kernel) -- harness numbers against the old
# form, parity tests that go through it, bench lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
bash scripts/exp/xerr_run.sh l "0 26 50 0 26 50" > /dev/null 2>&1
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py -x -q -m gpu 2>&1 | grep -v "$F" | tail -5 > gpurun_out/r05/l_tests.log
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | grep -v "$F" | tail -5 >> gpurun_out/r05/l_tests.log
B="python bench.py --cpu-iters 0 --others 0 --call 0"
$B --steps 20 --warmup 5 > gpurun_out/r05/l_bench_steps20.json 2> gpurun_out/r05/l_bench_steps20.err
$B > gpurun_out/r05/l_bench.json 2> gpurun_out/r05/l_bench.err
$B --trace 1 > gpurun_out/r05/l_bench_trace1.json 2> gpurun_out/r05/l_bench_trace1.err
cat gpurun_out/r05/xerr_exp_l.log | grep variant
cat gpurun_out/r05/l_tests.log
python - <<'PY'
import json
for f in ("l_bench_steps20", "l_bench", "l_bench_trace1"):
    try:
        d = json.load(open(f"gpurun_out/r05/{f}.json"))
        print(f, round(d["value"], 1), [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["phases_ms"].items() if k != "note"}, "mse", d["final_mse"], d.get("mse_check") and d["mse_check"]["rel_diff"])
    except Exception as e:
        print(f, "ERR", e)
PY
