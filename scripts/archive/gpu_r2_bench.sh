#!/bin/bash
# One gpurun call: bench lines for configs 2, 3, 5 + rocprofv3 kernel stats for each (outputs under gpurun_out/r02/).
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
TAG=${1:-v1}
for cfg in 2 3 5; do
  steps=40; [ $cfg != 2 ] && steps=10
  timeout 900 python bench.py --config $cfg --steps $steps --warmup 2 > gpurun_out/r02/${TAG}_cfg${cfg}_bench.json 2> gpurun_out/r02/${TAG}_cfg${cfg}_bench.err; echo "cfg $cfg bench exit=$?"
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r02/${TAG}_cfg${cfg}_bench.json"))
    print("cfg", $cfg, "it/s", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 3), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3),
          {k: (round(v["ms_per_launch"], 3) if v["ms_per_launch"] else None) for k, v in d["kernels"].items()}, "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"], 3),
          "mse_check", d["mse_check"] and d["mse_check"]["rel_diff"])
except Exception as e:
    print("cfg", $cfg, "parse failed", e)
PY
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof_${TAG}_cfg${cfg} -o p -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps $steps --warmup 2 --cpu-iters 0 --repeats 1 > $GRAFT_REPO_ROOT/gpurun_out/r02/${TAG}_cfg${cfg}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02/${TAG}_cfg${cfg}_prof.err; echo "rocprof exit=$?")
  f=$(find gpurun_out/r02/prof_${TAG}_cfg${cfg} -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r02/${TAG}_cfg${cfg}_kernel_stats.csv && head -8 "$f" | cut -c1-160
  rm -rf gpurun_out/r02/prof_${TAG}_cfg${cfg}
done
