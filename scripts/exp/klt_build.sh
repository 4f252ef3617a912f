#!/bin/bash
# builds the klt_exp binaries (scripts/exp/klt_exp.hip): plain, pinned, and the ablations of the pinned kernel
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value"
/opt/rocm/bin/hipcc $F -o klt_exp klt_exp.hip &
/opt/rocm/bin/hipcc $F -DKLT_PIN=1 -o klt_exp_pin klt_exp.hip &
for e in 1 2 4 8 5 9 12 13 15; do /opt/rocm/bin/hipcc $F -DKLT_PIN=1 -DKLT_EXP=$e -o klt_exp_pin_e$e klt_exp.hip & done
wait
ls -la klt_exp klt_exp_pin*
