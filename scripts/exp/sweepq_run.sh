#!/bin/bash
# GPU-side driver of scripts/exp/sweepq_exp (see sweepq_exp.hip)
cd "$(dirname "$0")"
E=./sweepq_exp
for args in "10000 50 50" "20000 50 50" "1280 50 50" "2560 50 50"; do REL_TOL=-1 timeout 60 $E $args; done
for args in "10000 50 50" "20000 50 50"; do timeout 60 $E $args; done
for k in 12 20 32 48 50 64; do timeout 60 $E 333 $k 50; MASK=1 timeout 60 $E 333 $k 50; done
REL_TOL=1e-3 timeout 60 ./sweepq_exp 777 50 50
REL_TOL=1e-3 MASK=1 timeout 60 ./sweepq_exp 777 50 50
GRAM=1 REL_TOL=-1 timeout 60 $E 20000 50 50
