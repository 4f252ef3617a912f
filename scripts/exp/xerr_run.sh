#!/bin/bash
# runs the xerr_exp variants on the GPU box (binary built in the container by hipcc, travels with the snapshot)
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out/r05
OUT=../../gpurun_out/r05/xerr_exp_${1:-a}.log
: > $OUT
for v in ${2:-0 1 3 4 5 6 0 5 13 14 15 16}; do timeout 120 ./xerr_exp 20000 10000 $v 20 >> $OUT 2>&1; done
cat $OUT
