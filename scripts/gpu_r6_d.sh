#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
PMC= $R/scripts/gpu_prof.sh e_cfg5 5 f32 8 | cut -c1-200 | head -6
cp $R/gpurun_out/prof/e_cfg5_kernel_stats.csv $O/
