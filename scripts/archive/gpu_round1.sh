#!/bin/bash
# One gpurun call: GPU tests, a bench line, and a rocprofv3 kernel-trace summary of the same bench command.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 40 --warmup 4 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit=$?"
cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --cpu-iters 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err; echo "rocprof exit=$?")
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
