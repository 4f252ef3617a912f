#!/bin/bash
# GPU-side driver of scripts/exp/sweepf_exp (see sweepf_exp.hip): the fp32-chain sweep (k_sweep_f.h) next to the fp64-chain sweep
cd "$(dirname "$0")"
E=./sweepf_exp
for w in 4 8; do
  for args in "10000 50 50" "20000 50 50" "1280 50 50" "40000 50 50"; do TIMING=1 NW=$w REL_TOL=-1 timeout 60 $E $args; done
  NW=$w timeout 60 $E 10000 50 50
  WARM=1 NW=$w REL_TOL=-1 timeout 120 $E 4000 50 50
done
if [ "$1" == "full" ]; then
for w in 4 8; do
for k in 1 12 16 20 32 48 50 64; do NW=$w timeout 60 $E 333 $k 50; NW=$w MASK=1 timeout 60 $E 333 $k 50; done
NW=$w REL_TOL=1e-3 timeout 60 $E 777 50 50
NW=$w REL_TOL=1e-3 MASK=1 timeout 60 $E 777 50 50
NW=$w GRAM=1 SLABS=3 REL_TOL=-1 timeout 60 $E 20000 50 50
done
fi
