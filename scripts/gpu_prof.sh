#!/bin/bash
# rocprofv3 passes over the bench command: kernel-trace stats, then HBM byte counters (separate passes).
set -x
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/$TAG -o trace -- python $R/bench.py --steps 20 --warmup 2 --cpu-iters 0 > $R/gpurun_out/prof/${TAG}_bench.json 2> $R/gpurun_out/prof/${TAG}_trace.err; echo "trace exit=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/$TAG -o fetch -- python $R/bench.py --steps 6 --warmup 1 --cpu-iters 0 > /dev/null 2> $R/gpurun_out/prof/${TAG}_fetch.err; echo "fetch exit=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/$TAG -o write -- python $R/bench.py --steps 6 --warmup 1 --cpu-iters 0 > /dev/null 2> $R/gpurun_out/prof/${TAG}_write.err; echo "write exit=$?"
cd $R
ls -la gpurun_out/prof/$TAG
head -30 gpurun_out/prof/$TAG/*kernel_stats.csv
python scripts/pmc_summary.py gpurun_out/prof/$TAG || true
# keep the merged output small: the per-dispatch traces are large
find gpurun_out/prof/$TAG -name "*kernel_trace.csv" -size +2M -delete
find gpurun_out/prof/$TAG -name "*.db" -delete
