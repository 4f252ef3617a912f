#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
echo "== overlap probe, 3-stage cross product"; timeout 600 python scripts/gpu_overlap_probe.py 2>&1 | tail -6
echo "== overlap probe, 2-stage cross product"; NNLM_EXP_XPROD_NBUF2=1 timeout 600 python scripts/gpu_overlap_probe.py 2>&1 | tail -6
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
