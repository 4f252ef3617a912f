#!/bin/bash
# round 5, call M: fp64 cross product with wavefronts 4..7 issuing behind their MFMA phase -- strict-mode parity tests and bench lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
(cd scripts/exp && timeout 300 ./xprod64_exp 1) > gpurun_out/r05/m_xprod64_exp.log 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | grep -v "$F" | tail -5 > gpurun_out/r05/m_tests.log
B="python bench.py --cpu-iters 0 --others 0 --call 0"
$B --precision f64 > gpurun_out/r05/m_bench_f64.json 2> gpurun_out/r05/m_bench_f64.err
$B --precision f64 --steps 20 --warmup 5 > gpurun_out/r05/m_bench_f64_steps20.json 2> gpurun_out/r05/m_bench_f64_steps20.err
cat gpurun_out/r05/m_xprod64_exp.log gpurun_out/r05/m_tests.log
python - <<'PY'
import json
for f in ("m_bench_f64", "m_bench_f64_steps20"):
    try:
        d = json.load(open(f"gpurun_out/r05/{f}.json"))
        print(f, round(d["value"], 1), [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["phases_ms"].items() if k != "note"}, "mse", d["final_mse"])
    except Exception as e:
        print(f, "ERR", e)
PY
