#!/bin/bash
cd "$(dirname "$0")/exp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lane_exp lane_exp.hip 2>/dev/null
mkdir -p ../../gpurun_out/r06
timeout 120 /tmp/lane_exp | tee ../../gpurun_out/r06/lane_exp.log
