#!/bin/bash
# deep runs of the closing build: 300-seed fuzz, 60 seeds with the device made to look like 3 and 7 CUs, the 40000-column parity test
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
F() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; }
python -m pytest tests/test_gpu_wrap.py -x -q -m gpu -k "forty" 2>&1 | F | tail -3 > gpurun_out/r06/k_tests_wrap.log
NNLM_FUZZ_SEEDS=300 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | F | tail -3 > gpurun_out/r06/k_fuzz300.log
NNLM_TEST_CUS=3 NNLM_FUZZ_SEEDS=60 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | F | tail -3 > gpurun_out/r06/k_fuzz60_cus3.log
NNLM_TEST_CUS=7 NNLM_FUZZ_SEEDS=60 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | F | tail -3 > gpurun_out/r06/k_fuzz60_cus7.log
for f in gpurun_out/r06/k_*.log; do echo "== $f"; tail -n 3 "$f"; done
