#!/bin/bash
# One gpurun call: rocprofv3 kernel stats of one bench configuration (and, with PMC=1, the HBM byte counters in separate passes).
# usage: scripts/gpu_prof.sh TAG CONFIG PRECISION [STEPS] [extra bench.py arguments]      e.g.  gpu_prof.sh r06_cfg2 2 f32 20
# -> gpurun_out/prof/${TAG}_bench_under_rocprof.json, ${TAG}_kernel_stats.csv (+ ${TAG}_pmc_hbm_summary.txt); copy what is to be kept into profiles/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; CFG=$2; PREC=$3; STEPS=${4:-10}; shift 4
OUT=$R/gpurun_out/prof
mkdir -p $OUT
BENCH="python $R/bench.py --config $CFG --precision $PREC --steps $STEPS --warmup 2 --cpu-iters 0 --repeats 1 --others 0 --call 0 $*"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG} -o p -- $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_prof.err; echo "rocprof exit=$?")
f=$(find $OUT/prof_${TAG} -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_kernel_stats.csv && head -16 "$f" | cut -c1-170
rm -rf $OUT/prof_${TAG}
if [ "$PMC" == "1" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$c -o p -- $BENCH --steps 6 > /dev/null 2> $OUT/${TAG}_pmc_$c.err; echo "pmc $c exit=$?")
  done
  python $R/scripts/pmc_summary.py $OUT/pmc_${TAG}_FETCH_SIZE $OUT/pmc_${TAG}_WRITE_SIZE > $OUT/${TAG}_pmc_hbm_summary.txt 2>&1 || true
  tail -20 $OUT/${TAG}_pmc_hbm_summary.txt
  rm -rf $OUT/pmc_${TAG}_FETCH_SIZE $OUT/pmc_${TAG}_WRITE_SIZE
fi
