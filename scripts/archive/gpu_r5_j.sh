#!/bin/bash
# round 5, call J: the error block as a split-fp16 kernel beside the sweep (err_mode 1) against the fused cross product (err_mode 0)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
B="python bench.py --steps 20 --warmup 5 --cpu-iters 2 --others 0 --call 0"
for mode in 0 1; do
  NNLM_EXP_ERR_MODE=$mode $B > gpurun_out/r05/j_bench_mode$mode.json 2> gpurun_out/r05/j_bench_mode$mode.err
done
for S in 2 8 16; do
  NNLM_EXP_ERR_MODE=1 NNLM_EXP_ERR16_S=$S $B --cpu-iters 0 > gpurun_out/r05/j_bench_mode1_s$S.json 2>/dev/null
done
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config2 or twenty or driver" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r05/j_tests.log
cat gpurun_out/r05/j_tests.log
python - <<'PY'
import json
for f in ("j_bench_mode0", "j_bench_mode1", "j_bench_mode1_s2", "j_bench_mode1_s8", "j_bench_mode1_s16"):
    try:
        d = json.load(open(f"gpurun_out/r05/{f}.json"))
        print(f, round(d["value"], 1), [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: round(v, 4) for k, v in d["phases_ms"].items() if k != "note"}, "mse", d["final_mse"], d.get("mse_check") and d["mse_check"]["rel_diff"])
    except Exception as e:
        print(f, "ERR", e)
PY
