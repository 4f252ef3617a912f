#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
B="python $R/bench.py --steps 8 --warmup 2 --cpu-iters 0 --repeats 2 --others 0 --call 0 --config 5"
for p in f32; do
  timeout 600 $B --precision $p > $O/st_$p.json 2> $O/st_$p.err
  python -c "import json; d=json.load(open('$O/st_$p.json')); print('$p', 'ms/step', round(d['ms_per_step'],4), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if v['ms_per_launch']}, 'mse', d.get('final_mse'))"
done 2>&1 | tee $O/colsolve_step.log
(cd $R && timeout 1500 python -m pytest tests -m gpu -x -q -k "missing or na or fuzz or config5 or edges" > $O/d_gputests.log 2>&1; echo "gpu tests exit=$?"; grep -n "passed\|failed" $O/d_gputests.log | tail -3)
PMC= scripts/gpu_prof.sh d_cfg5 5 f32 8 | cut -c1-160 | head -5
