#!/bin/bash
# round 5, call P: two-part split-K plan of the dense split-fp16 cross products -- parity tests, bench lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_boundary.py tests/test_gpu_wrap.py -x -q -m gpu 2>&1 | grep -v "$F" | tail -15 > gpurun_out/r05/p_tests.log
B="python bench.py --cpu-iters 0 --others 0 --call 0"
$B --steps 20 --warmup 5 > gpurun_out/r05/p_bench_steps20.json 2> gpurun_out/r05/p_bench_steps20.err
$B > gpurun_out/r05/p_bench.json 2> gpurun_out/r05/p_bench.err
cat gpurun_out/r05/p_tests.log
python - <<'PY'
import json
for f in ("p_bench_steps20", "p_bench"):
    try:
        d = json.load(open(f"gpurun_out/r05/{f}.json"))
        print(f, round(d["value"], 1), round(d["ms_per_step"], 4), [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["phases_ms"].items() if k != "note"}, "mse", d["final_mse"])
    except Exception as e:
        print(f, "ERR", e)
PY
