#!/bin/bash
# One gpurun call: the GPU test-suite (args: pytest selection), output under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout ${TEST_TIMEOUT:-1500} python -m pytest "$@" -m gpu -q --no-header --tb=short --durations=25 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit=$?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
