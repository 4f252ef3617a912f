#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_wrap.py tests/test_gpu_edges.py -q -x 2>&1 | tail -6
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_wrap.py --deselect tests/test_gpu_edges.py 2>&1 | tail -6
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_wrap2.json 2> $O/bench_wrap2.err; echo "bench exit=$?"
python - <<PY
import json
d = json.load(open("$O/bench_wrap2.json"))
print("it/s", round(d["value"], 1), "repeats", [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: round(v, 4) for k, v in d["phases_ms"].items() if k != "note"}, "mse", d["final_mse"])
print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["bound"])
PY
