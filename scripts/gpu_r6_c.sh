#!/bin/bash
# config 5 with the product's solver, then the GPU suite and a 100-seed fuzz
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
B="python $R/bench.py --steps 8 --warmup 2 --cpu-iters 0 --repeats 2 --others 0 --call 0 --config 5"
for p in f32 f64; do
  timeout 600 $B --precision $p > $O/fin5_$p.json 2> $O/fin5_$p.err
  python -c "import json; d=json.load(open('$O/fin5_$p.json')); print('$p', 'ms/step', round(d['ms_per_step'],4), {k: round(v['ms_per_launch'],4) for k,v in d['kernels'].items() if v['ms_per_launch']}, 'mse', d.get('final_mse'))"
done 2>&1 | tee $O/fin5.log
(cd $R && timeout 1500 python -m pytest tests -m gpu -x -q > $O/f_gputests.log 2>&1; echo "gpu tests exit=$?"; grep -n "passed\|failed\|Error" $O/f_gputests.log | tail -5)
(cd $R && NNLM_FUZZ_SEEDS=100 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 | tee $O/f_fuzz100.log)
