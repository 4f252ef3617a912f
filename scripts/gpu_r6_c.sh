#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
B="python $R/bench.py --steps 8 --warmup 2 --cpu-iters 0 --repeats 2 --others 0 --call 0 --config 5 --precision f32"
for v in upper full upper full; do
  [ $v == full ] && export NNLM_EXP_NA_FULL=1 || unset NNLM_EXP_NA_FULL
  timeout 600 $B > $O/nf_$v.json 2> $O/nf_$v.err
  python -c "import json; d=json.load(open('$O/nf_$v.json')); print('store $v', 'ms/step', round(d['ms_per_step'],4), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if v['ms_per_launch']}, 'mse', d.get('final_mse'))"
done 2>&1 | tee $O/na_full_vec_ab.log
