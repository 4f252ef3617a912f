#!/bin/bash
# GPU-side driver of scripts/exp/sweepq_exp (see sweepq_exp.hip).  Q4=all: all columns as 4-column wavefronts; Q4=0: none; default: the product's split
cd "$(dirname "$0")"
E=./sweepq_exp
for q4 in 0 split all; do
  for args in "10000 50 50" "20000 50 50" "1250 50 50" "2500 50 50" "4096 50 50"; do Q4=$q4 REL_TOL=-1 timeout 60 $E $args; done
done
for args in "10000 50 50" "20000 50 50"; do timeout 60 $E $args; STRICT=1 timeout 60 $E $args; done
for k in 1 12 16 20 32 48 50 64; do Q4=all timeout 60 $E 333 $k 50; Q4=all STRICT=1 timeout 60 $E 333 $k 50; MASK=1 timeout 60 $E 333 $k 50; done
Q4=all REL_TOL=1e-3 timeout 60 ./sweepq_exp 777 50 50
Q4=all STRICT=1 REL_TOL=1e-3 timeout 60 ./sweepq_exp 777 50 50
REL_TOL=1e-3 MASK=1 timeout 60 ./sweepq_exp 777 50 50
GRAM=1 REL_TOL=-1 timeout 60 $E 20000 50 50
GRAM=1 SLABS=3 REL_TOL=-1 timeout 60 $E 20000 50 50
