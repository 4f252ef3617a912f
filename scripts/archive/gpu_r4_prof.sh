#!/bin/bash
# One gpurun call: rocprofv3 kernel stats of one bench configuration.  usage: gpu_r4_prof.sh TAG CONFIG PRECISION STEPS
# -> gpurun_out/r04/${TAG}_bench_under_rocprof.json, ${TAG}_kernel_stats.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; CFG=$2; PREC=$3; STEPS=${4:-10}
mkdir -p $R/gpurun_out/r04
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04/prof_${TAG} -o p -- python $R/bench.py --config $CFG --precision $PREC --steps $STEPS --warmup 2 --cpu-iters 0 --repeats 1 --others 0 > $R/gpurun_out/r04/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/r04/${TAG}_prof.err; echo "rocprof exit=$?")
f=$(find $R/gpurun_out/r04/prof_${TAG} -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r04/${TAG}_kernel_stats.csv && head -14 "$f" | cut -c1-170
rm -rf $R/gpurun_out/r04/prof_${TAG}
