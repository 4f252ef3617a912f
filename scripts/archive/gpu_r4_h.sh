#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for sz in 20000,10000,50 20000,8192,50 16384,10000,50 16384,8192,50; do
  timeout 300 python bench.py --size $sz --steps 20 --warmup 5 --cpu-iters 0 --others 0 --repeats 2 > /tmp/b.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
ra = d["roofline_all"]
print("$sz", "ms/step", [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: (round(v["ms_per_launch"], 4), round(v["achieved"], 0)) for k, v in ra.items()})
PY
done
