#!/bin/bash
# round 4, closing call: the whole GPU suite, the driver's bench line, smoke()
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|rror" | head
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/cfg2_bench.json 2> gpurun_out/r04/cfg2_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/r04/cfg2_bench.json"))
print(round(d["value"], 1), d["repeats"]["ms_per_step"], "upload_s", d["upload_and_prep_s"])
print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
PY
python -c "import __graft_entry__ as g; g.smoke()"
