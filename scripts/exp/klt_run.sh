#!/bin/bash
# GPU-side driver of the klt_exp binaries: both half-step shapes of config 3
cd "$(dirname "$0")"
for shape in "20000 10000" "10000 20000"; do
  echo "== shape $shape (contraction, columns), Lee (method 4)"
  timeout 120 ./klt_exp $shape 50 4 0 6
  timeout 120 ./klt_exp_pin $shape 50 4 0 6
  timeout 120 ./klt_exp_pin $shape 50 4 2 6
  for e in 1 2 4 8 5 9 12 13 15; do timeout 120 ./klt_exp_pin_e$e $shape 50 4 0 4; done
done
echo "== SCD-KL (method 3), 2 sweeps"
timeout 120 ./klt_exp_pin 20000 10000 50 3 0 4 2
timeout 120 ./klt_exp_pin 20000 10000 50 3 2 4 2
timeout 120 ./klt_exp_pin 10000 20000 50 3 0 4 2
timeout 120 ./klt_exp_pin 10000 20000 50 3 2 4 2
echo "== small / ragged"
timeout 60 ./klt_exp_pin 5000 3001 50 4 0 3 3
timeout 60 ./klt_exp_pin 5000 3001 50 4 2 3 3
timeout 60 ./klt_exp_pin 15000 777 13 4 0 3 2
timeout 60 ./klt_exp_pin 15000 777 13 4 2 3 2
