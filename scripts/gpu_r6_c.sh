#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --cpu-iters 0 --repeats 3 --others 0 --call 0"
for v in mfma row mfma row; do
  [ $v == row ] && export NNLM_EXP_SWEEP_ROW=1 || unset NNLM_EXP_SWEEP_ROW
  timeout 600 $B > $O/sr_$v.json 2> $O/sr_$v.err
  python -c "import json; d=json.load(open('$O/sr_$v.json')); print('sweep $v', 'it/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if v['ms_per_launch']}, 'mse', d.get('final_mse'))"
done 2>&1 | tee $O/sweep_row_ab.log
export NNLM_EXP_SWEEP_ROW=1
(cd $R && timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -m gpu -x -q > $O/g_gputests.log 2>&1; echo "gpu tests (row form) exit=$?"; grep -n "passed\|failed\|Error\|assert" $O/g_gputests.log | tail -8)
