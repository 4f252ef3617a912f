#!/bin/bash
# round 4: the whole GPU suite + first-region probes of the bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
for spec in "w5 --warmup 5" "w50 --warmup 50" "w5b --warmup 5"; do
  set -- $spec
  tag=$1; shift
  timeout 600 python bench.py --steps 20 "$@" --cpu-iters 0 --others 0 --repeats 4 > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
d = json.load(open("$O/bench_$tag.json"))
print("$tag", "it/s", round(d["value"], 1), "repeats", [round(x, 4) for x in d["repeats"]["ms_per_step"]])
PY
done
