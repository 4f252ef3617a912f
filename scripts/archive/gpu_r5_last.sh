#!/bin/bash
# round 5, closing call: the whole GPU suite, the driver's bench line, smoke()
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
rm -f gpurun_out/parity_report.jsonl
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|rror" | head
cp gpurun_out/parity_report.jsonl gpurun_out/r05/parity_report.jsonl 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/last_bench.json 2> gpurun_out/r05/last_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/r05/last_bench.json"))
print(round(d["value"], 1), d["repeats"]["ms_per_step"], "upload_s", d["upload_and_prep_s"])
print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
print({k: v.get("ms_per_step") for k, v in d["other_configs"].items()})
PY
python -c "import __graft_entry__ as g; g.smoke()"
