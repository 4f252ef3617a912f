#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fuzz.py 2>&1 | tail -6
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_img.json 2> $O/bench_img.err; echo "bench exit=$?"
python - <<PY
import json
d = json.load(open("$O/bench_img.json"))
print("it/s", round(d["value"], 1), "repeats", [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: round(v, 4) for k, v in d["phases_ms"].items() if k != "note"}, "mse", d["final_mse"])
for k, v in (d.get("other_configs") or {}).items():
    print("  ", k, v.get("ms_per_step"), v.get("kernels_ms_per_launch"), v.get("error"))
PY
