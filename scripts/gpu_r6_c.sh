#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
B="python $R/bench.py --steps 8 --warmup 2 --cpu-iters 0 --repeats 2 --others 0 --call 0 --config 5"
for p in f32 f64; do
  timeout 600 $B --precision $p > $O/up2_$p.json 2> $O/up2_$p.err
  python -c "import json; d=json.load(open('$O/up2_$p.json')); print('$p', 'ms/step', round(d['ms_per_step'],4), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if v['ms_per_launch']}, 'mse', d.get('final_mse'))"
done 2>&1 | tee $O/upper_gram_ab2.log
(cd $R && timeout 1500 python -m pytest tests -m gpu -x -q > $O/c_gputests.log 2>&1; echo "gpu tests exit=$?"; grep -n "passed\|failed" $O/c_gputests.log | tail -3)
