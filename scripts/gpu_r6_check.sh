#!/bin/bash
# One gpurun call: the GPU suite, the driver-protocol bench line and (optionally) the rocprofv3 summaries of the build in the tree.
# usage: scripts/gpu_r6_check.sh TAG [what...]     what = tests bench stats pmc others (default: tests bench stats)
# -> gpurun_out/r06/${TAG}_*; copy what is to be kept into profiles/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/r06
mkdir -p $O
WHAT="${@:-tests bench stats}"
if [[ $WHAT == *tests* ]]; then
  (cd $R && timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_gputests.log 2>&1; echo "gpu tests exit=$?"; tail -3 $O/${TAG}_gputests.log)
fi
if [[ $WHAT == *bench* ]]; then
  timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_cfg2_bench.json 2> $O/${TAG}_cfg2_bench.err; echo "bench exit=$?"
  python - <<PY
import json
d = json.load(open("$O/${TAG}_cfg2_bench.json"))
print("cfg2 it/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), {k: (round(v["ms_per_launch"], 4) if v["ms_per_launch"] else None) for k, v in d["kernels"].items()})
print("roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
for k, v in (d.get("other_configs") or {}).items():
    print(k, v.get("ms_per_step"), v.get("dominant"), v.get("error"))
print("cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"], "mse_check", d["mse_check"])
PY
fi
if [[ $WHAT == *stats* ]]; then
  for spec in "cfg2 2 f32 20" "cfg3 3 f32 6" "cfg5 5 f32 8" "f64_cfg2 2 f64 8" "f64_cfg3 3 f64 3" "f64_cfg5 5 f64 4"; do
    set -- $spec
    [[ $WHAT == *cfg2only* && $1 != cfg2 ]] && continue
    PMC=$([[ $WHAT == *pmc* ]] && echo 1) $R/scripts/gpu_prof.sh ${TAG}_$1 $2 $3 $4 | cut -c1-150
  done
  cp $R/gpurun_out/prof/${TAG}_* $O/ 2>/dev/null
fi
