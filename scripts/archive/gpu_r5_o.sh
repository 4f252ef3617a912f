#!/bin/bash
# round 5, call O: the NA flow's error block inside the fused cross product (xprod16_err_kernel<NKQ, true>) -- fused against separate
# error values on small edge cases, parity tests, config 5 and config 2 bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
NNLM_PRECISION=f32 python scripts/gpu_dbg_na_err.py 2>&1 | grep -v "$F" > gpurun_out/r05/o_na_err.log
bash scripts/exp/xerr_run.sh o "0 50 0 50" > /dev/null 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | grep -v "$F" | tail -15 > gpurun_out/r05/o_tests.log
B="python bench.py --cpu-iters 0 --others 0 --call 0"
$B --config 5 > gpurun_out/r05/o_bench_cfg5.json 2> gpurun_out/r05/o_bench_cfg5.err
$B --steps 20 --warmup 5 > gpurun_out/r05/o_bench_cfg2_steps20.json 2> gpurun_out/r05/o_bench_cfg2_steps20.err
cat gpurun_out/r05/o_na_err.log; grep variant gpurun_out/r05/xerr_exp_o.log; cat gpurun_out/r05/o_tests.log
python - <<'PY'
import json
for f in ("o_bench_cfg5", "o_bench_cfg2_steps20"):
    try:
        d = json.load(open(f"gpurun_out/r05/{f}.json"))
        print(f, round(d["value"], 1), round(d["ms_per_step"], 4), [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["phases_ms"].items() if k != "note"}, "mse", d["final_mse"])
    except Exception as e:
        print(f, "ERR", e)
PY
