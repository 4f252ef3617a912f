#!/bin/bash
# Round-2 evidence pass (one gpurun call): bench lines + rocprofv3 kernel stats for configs 2 / 3 / 5, HBM-traffic PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, no tracing) for config 2 and 3, P-core protocol line.  Outputs: gpurun_out/r02/<tag>_*.
TAG=${1:-final}
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
bash scripts/gpu_r2_bench.sh $TAG
timeout 600 python bench.py --protocol core --cpu-iters 0 > gpurun_out/r02/${TAG}_cfg2_pcore_bench.json 2>/dev/null
for cfg in 2 3 5; do
  : > gpurun_out/r02/${TAG}_cfg${cfg}_pmc_hbm_summary.txt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/pmc_$ctr -o p -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 6 --warmup 1 --cpu-iters 0 --repeats 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r02/${TAG}_cfg${cfg}_pmc_$ctr.err; echo "pmc cfg$cfg $ctr exit=$?")
    python scripts/pmc_summary.py "$(dirname "$(find gpurun_out/r02/pmc_$ctr -name '*counter_collection.csv' | head -1)")" | grep -E "xprod|sweep|errors_|kl_tile|wh_store|na_gram|colsolve" >> gpurun_out/r02/${TAG}_cfg${cfg}_pmc_hbm_summary.txt
    rm -rf gpurun_out/r02/pmc_$ctr
  done
done
head -30 gpurun_out/r02/${TAG}_cfg2_pmc_hbm_summary.txt
python - <<PY
import json
d = json.load(open("gpurun_out/r02/${TAG}_cfg2_pcore_bench.json")); print("P-core", round(d["value"], 1), "it/s", d["repeats"]["ms_per_step"])
PY
