#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
B="python $R/bench.py --steps 8 --warmup 2 --cpu-iters 0 --repeats 2 --others 0 --call 0 --config 5 --precision f32"
for v in 1 2; do
  timeout 600 $B > $O/w1_$v.json 2> $O/w1_$v.err
  python -c "import json; d=json.load(open('$O/w1_$v.json')); print('one-wave workgroups', 'ms/step', round(d['ms_per_step'],4), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if v['ms_per_launch']}, 'mse', d.get('final_mse'))"
done 2>&1 | tee $O/colsolve_row_w1.log
PMC= $R/scripts/gpu_prof.sh f_cfg5 5 f32 8 | grep "colsolve\|na_gram" | cut -c1-60,150-260
(cd $R && timeout 1500 python -m pytest tests -m gpu -x -q -k "missing or na or fuzz or config5 or edges or nsclc or boundary" > $O/h_gputests.log 2>&1; echo "gpu tests exit=$?"; grep -n "passed\|failed" $O/h_gputests.log | tail -3)
