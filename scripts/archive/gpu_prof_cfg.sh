#!/bin/bash
# rocprofv3 kernel stats over the full-size KL / NA configurations (scripts/gpu_cfg35.py, no oracle check)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp
CHECK=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/cfg35 -o trace -- python $R/scripts/gpu_cfg35.py > $R/gpurun_out/prof/cfg35.log 2>&1
cd $R
cat gpurun_out/prof/cfg35.log | grep cfg
head -14 gpurun_out/prof/cfg35/*kernel_stats.csv | cut -c1-200
find gpurun_out/prof/cfg35 -name "*kernel_trace.csv" -delete
