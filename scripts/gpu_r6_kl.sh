#!/bin/bash
# A/B of the KL tile kernel's starting states (in-kernel against the wh_store GEMM), strict config 5 / 3b timings
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
B="python $R/bench.py --steps 6 --warmup 2 --cpu-iters 0 --repeats 2 --others 0 --call 0"
for v in own gemm; do
  [ $v == gemm ] && export NNLM_KL_INIT_GEMM=1 || unset NNLM_KL_INIT_GEMM
  timeout 600 $B --config 3 --precision f32 > $O/kl_${v}_cfg3.json 2> $O/kl_${v}_cfg3.err; echo "cfg3 $v exit=$?"
  python -c "import json; d=json.load(open('$O/kl_${v}_cfg3.json')); print('$v', 'ms/step', round(d['ms_per_step'],4), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if v['ms_per_launch']})"
done
unset NNLM_KL_INIT_GEMM
timeout 600 $B --config 5 --precision f64 > $O/f64_cfg5.json 2> $O/f64_cfg5.err; python -c "import json; d=json.load(open('$O/f64_cfg5.json')); print('f64 cfg5 ms/step', round(d['ms_per_step'],4), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if v['ms_per_launch']})"
timeout 600 $B --config 3 --precision f64 --steps 3 > $O/f64_cfg3.json 2> $O/f64_cfg3.err; python -c "import json; d=json.load(open('$O/f64_cfg3.json')); print('f64 cfg3 ms/step', round(d['ms_per_step'],4), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if v['ms_per_launch']})"
(cd $R && timeout 1500 python -m pytest tests -m gpu -x -q > $O/kl_gputests.log 2>&1; echo "gpu tests exit=$?"; grep -n "passed\|failed" $O/kl_gputests.log | tail -3)
