#!/bin/bash
cd "$(dirname "$0")"
for shape in "20000 10000" "10000 20000"; do
  echo "== shape $shape: staggered two-group kernel (variant 2) against variant 0"
  timeout 120 ./klt_exp_v7 $shape 50 4 0 6
  timeout 120 ./klt_exp_v7 $shape 50 4 2 6
done
timeout 120 ./klt_exp_v7 20000 10000 50 3 2 4 2
timeout 120 ./klt_exp_v7 5000 3001 50 4 2 3 3
