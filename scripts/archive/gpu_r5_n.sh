#!/bin/bash
# round 5, call N: NA flow (config 5) -- the error block of a trace iteration beside the Gram / solver kernels instead of beside the cross product
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
B="python bench.py --config 5 --cpu-iters 0 --others 0 --call 0"
for m in 0 1 0 1; do
  NNLM_EXP_ERRLATE=$m $B > gpurun_out/r05/n_bench_cfg5_late$m.json 2> gpurun_out/r05/n_bench_cfg5_late$m.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r05/n_bench_cfg5_late$m.json"))
print("late=$m", round(d["ms_per_step"], 4), [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["phases_ms"].items() if k != "note"}, "mse", d["final_mse"])
PY
done
