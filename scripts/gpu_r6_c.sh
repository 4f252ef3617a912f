#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/final_gputests.log 2>&1; echo "gpu tests exit=$?"; grep -n "passed\|failed\|FAILED" $O/final_gputests.log | tail -8
grep "f32_early_stop" gpurun_out/parity_report.jsonl | tail -6
