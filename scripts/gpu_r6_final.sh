#!/bin/bash
# last call of the round: GPU suite, smoke, the driver's bench command
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/final_gputests.log 2>&1; echo "gpu tests exit=$?"; grep -n "passed\|failed" $O/final_gputests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err; echo "bench exit=$?"
python -c "import json; d=json.load(open('$O/final_bench.json')); print('it/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), {k: round(v['ms_per_step'],3) for k,v in d['other_configs'].items()})"
