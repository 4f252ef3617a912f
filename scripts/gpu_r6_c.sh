#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R && timeout 1500 python -m pytest tests -m gpu -x -q -k "kl or KL or mkl or config3" > $O/i_gputests.log 2>&1; echo "gpu tests exit=$?"; grep -n "passed\|failed\|Error" $O/i_gputests.log | tail -4
