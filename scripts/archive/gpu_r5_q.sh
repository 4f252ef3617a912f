#!/bin/bash
# round 5, call Q: A/B of the two-part split-K plan on one box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
B="python bench.py --cpu-iters 0 --others 0 --call 0 --steps 20 --warmup 5"
for m in "1 0.985" "0 0.985" "0 1.02" "1 0.985" "0 0.985" "0 1.02"; do
set -- $m
NNLM_EXP_PLAN2_OFF=$1 NNLM_EXP_PLAN2_THR=$2 $B > gpurun_out/r05/q_bench.json 2> gpurun_out/r05/q_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/r05/q_bench.json"))
print("off/thr=$m", round(d["value"], 1), [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["phases_ms"].items() if k != "note"})
PY
done
