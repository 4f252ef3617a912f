#!/bin/bash
# round 5, call T: 700-seed fuzz of the closing build (driver / c_nnlm / virtual ranks), and the F32 mode on the strict mode's hard cases
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
NNLM_FUZZ_SEEDS=700 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | grep -v "$F" | tail -4 > gpurun_out/r05/t_fuzz700.log
NNLM_FUZZ_F32_HARD=1 timeout 1200 python tests/fuzz_table.py 2>&1 | grep -v "$F" | tail -12 > gpurun_out/r05/t_fuzz_table_f32_hard.log
cat gpurun_out/r05/t_fuzz700.log gpurun_out/r05/t_fuzz_table_f32_hard.log
