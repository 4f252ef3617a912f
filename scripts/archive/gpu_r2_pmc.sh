#!/bin/bash
# PMC pass over one bench config: SQ busy/wait counters and HBM traffic of the dominant kernels (separate passes, no tracing).
# usage: gpu_r2_pmc.sh <tag> <config> [kernel regex]
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
TAG=$1; CFG=$2; KRE=${3:-.}
run() { # name, counters
  (cd /tmp && timeout 900 rocprofv3 --pmc $2 --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/pmc_$1 -o p -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 4 --warmup 1 --cpu-iters 0 --repeats 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r02/${TAG}_cfg${CFG}_pmc_$1.err; echo "pmc $1 exit=$?")
  f=$(find gpurun_out/r02/pmc_$1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scripts/pmc_summary.py "$(dirname "$f")" | grep -E "$KRE" >> gpurun_out/r02/${TAG}_cfg${CFG}_pmc_summary.txt
  rm -rf gpurun_out/r02/pmc_$1
}
: > gpurun_out/r02/${TAG}_cfg${CFG}_pmc_summary.txt
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
run sq2 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run tcc "TCC_HIT_sum TCC_MISS_sum"
cat gpurun_out/r02/${TAG}_cfg${CFG}_pmc_summary.txt
