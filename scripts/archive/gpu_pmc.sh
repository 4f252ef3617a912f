#!/bin/bash
# PMC pass over scripts/gpu_time.py: usage gpu_pmc.sh <tag> "<counters...>"
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p $R/gpurun_out/pmc
cd /tmp
ITERS=3 timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$TAG -o pmc -- python $R/scripts/gpu_time.py > $R/gpurun_out/pmc/${TAG}.log 2>&1
cd $R
python scripts/pmc_summary.py gpurun_out/pmc/$TAG | grep -E "xprod|sweep|errors_|gram_partial" 
find gpurun_out/pmc/$TAG -name "*kernel_trace.csv" -delete
