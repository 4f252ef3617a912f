#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== 3-stage"; timeout 600 python scripts/gpu_overlap_probe2.py 2>&1 | tail -4
echo "== 2-stage"; NNLM_EXP_XPROD_NBUF2=1 timeout 600 python scripts/gpu_overlap_probe2.py 2>&1 | tail -4
