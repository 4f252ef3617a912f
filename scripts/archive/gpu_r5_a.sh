#!/bin/bash
# round 5, call A: the API / bench changes on the GPU -- targeted tests, the default bench line (call metric, threads1), forced-comm line with --form both
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_wrap.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05/a_tests1.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "virtual or sharded or rccl" 2>&1 | tail -5 > gpurun_out/r05/a_tests2.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/a_bench.json 2> gpurun_out/r05/a_bench.err
NNLM_BENCH_FORCE_COMM=1 python bench.py --steps 20 --warmup 5 --cpu-iters 0 > gpurun_out/r05/a_bench_forced.json 2> gpurun_out/r05/a_bench_forced.err
python bench.py --gpus 2 --steps 4 --warmup 1 --size 6000,4000,50 --cpu-iters 0 --others 0 > gpurun_out/r05/a_bench_g2.json 2> gpurun_out/r05/a_bench_g2.err; echo "rc=$?" >> gpurun_out/r05/a_bench_g2.err
tail -3 gpurun_out/r05/a_tests1.log gpurun_out/r05/a_tests2.log
python - <<'PY'
import json
for f in ("a_bench", "a_bench_forced"):
    try:
        d = json.load(open(f"gpurun_out/r05/{f}.json"))
        print(f, d["value"], d["ms_per_step"], json.dumps(d.get("call")), json.dumps(d.get("forms")), json.dumps((d.get("cpu_baseline") or {}).get("threads1")))
        print({k: v.get("ms_per_step") for k, v in (d.get("other_configs") or {}).items()})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r05/a_bench_g2.err
