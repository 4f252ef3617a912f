#!/bin/bash
# round 5, call S: A/B of the memory-side-cache prefetch beside the sweep (NNLM_EXP_PREFETCH = stages per block, 0 = off)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
for w in 5 60; do
B="python bench.py --cpu-iters 0 --others 0 --call 0 --steps 20 --warmup $w"
for m in 0 16 0 16; do
NNLM_EXP_PREFETCH=$m $B > gpurun_out/r05/s_bench.json 2> gpurun_out/r05/s_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/r05/s_bench.json"))
print("warmup=$w prefetch=$m", round(d["value"], 1), [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["phases_ms"].items() if k != "note"})
PY
done
done
