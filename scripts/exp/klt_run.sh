#!/bin/bash
# GPU-side driver of the klt_exp binaries: both half-step shapes of config 3, step-loop arrangements, the two-workgroups-per-CU form (variant 3)
cd "$(dirname "$0")"
for shape in "20000 10000" "10000 20000"; do
  echo "== shape $shape (contraction, columns), Lee (method 4)"
  for v in 0 1 2 4 7; do timeout 120 ./klt_exp_v$v $shape 50 4 0 6; done
  timeout 120 ./klt_exp_v7 $shape 50 4 3 6
  timeout 120 ./klt_exp_v0 $shape 50 4 3 6
  timeout 120 ./klt_exp_v7e1 $shape 50 4 0 4
  timeout 120 ./klt_exp_v0t $shape 50 4 0 3
  timeout 120 ./klt_exp_v7t $shape 50 4 0 3
  timeout 120 ./klt_exp_v7t $shape 50 4 3 3
done
echo "== SCD-KL (method 3), 2 sweeps"
for shape in "20000 10000" "10000 20000"; do
  timeout 120 ./klt_exp_v0 $shape 50 3 0 4 2
  timeout 120 ./klt_exp_v7 $shape 50 3 0 4 2
  timeout 120 ./klt_exp_v7 $shape 50 3 3 4 2
done
echo "== small / ragged"
timeout 60 ./klt_exp_v7 5000 3001 50 4 0 3 3
timeout 60 ./klt_exp_v7 5000 3001 50 4 3 3 3
timeout 60 ./klt_exp_v0 15000 777 13 4 0 3 2
timeout 60 ./klt_exp_v7 15000 777 13 4 0 3 2
