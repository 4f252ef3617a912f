#!/bin/bash
# builds the klt_exp binaries (scripts/exp/klt_exp.hip): step-loop arrangements KLT_V = 0 .. 7 (pinned), timing builds
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DKLT_PIN=1"
for v in 0 1 2 4 7; do /opt/rocm/bin/hipcc $F -DKLT_V=$v -o klt_exp_v$v klt_exp.hip & done
/opt/rocm/bin/hipcc $F -DKLT_V=0 -DKLT_TIMING=1 -o klt_exp_v0t klt_exp.hip &
/opt/rocm/bin/hipcc $F -DKLT_V=7 -DKLT_TIMING=1 -o klt_exp_v7t klt_exp.hip &
/opt/rocm/bin/hipcc $F -DKLT_V=7 -DKLT_EXP=1 -o klt_exp_v7e1 klt_exp.hip &
wait
ls -la klt_exp_v*
