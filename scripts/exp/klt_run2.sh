#!/bin/bash
cd "$(dirname "$0")"
for shape in "20000 10000" "10000 20000"; do
  echo "== shape $shape: one vector pass per step (variant 5) against variant 0"
  timeout 120 ./klt_exp_v7 $shape 50 4 0 6
  timeout 120 ./klt_exp_v7 $shape 50 4 5 6
done
timeout 120 ./klt_exp_v7 20000 10000 50 3 0 4 2
timeout 120 ./klt_exp_v7 20000 10000 50 3 5 4 2
timeout 120 ./klt_exp_v7 10000 20000 50 3 5 4 2
timeout 120 ./klt_exp_v7 5000 3001 50 4 0 3 3
timeout 120 ./klt_exp_v7 5000 3001 50 4 5 3 3
timeout 60 ./klt_exp_v7 15000 777 13 4 5 3 2
