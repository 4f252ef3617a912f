#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
B="python $R/bench.py --steps 8 --warmup 2 --cpu-iters 0 --repeats 2 --others 0 --call 0 --config 5 --precision f32"
for v in 0 2 4 8 16 0; do
  export NNLM_EXP_NA_CHUNKS=$v
  timeout 600 $B > $O/ch_$v.json 2> $O/ch_$v.err
  python -c "import json; d=json.load(open('$O/ch_$v.json')); print('chunks $v', 'ms/step', round(d['ms_per_step'],4), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if v['ms_per_launch']}, 'mse', d.get('final_mse'))"
done 2>&1 | tee $O/na_chunks_ab.log
