#!/bin/bash
# round 5, call F: sweep generalisation (wrap tests incl. timing assertion, full-size tests), default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_wrap.py -x -q -m gpu -s 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r05/f_tests_wrap.log
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r05/f_tests_full.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/f_bench.json 2> gpurun_out/r05/f_bench.err
cat gpurun_out/r05/f_tests_wrap.log gpurun_out/r05/f_tests_full.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/f_bench.json"))
print(d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"], {k: round(v, 4) for k, v in d["phases_ms"].items() if k != "note"})
print({k: v.get("ms_per_step") for k, v in (d.get("other_configs") or {}).items()})
print(json.dumps(d["call"]))
PY
