#!/bin/bash
# round 4, first GPU call: the new tests, the driver's bench line, the forced-communicator line (phases_ms), first-region probe
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_edges.py tests/test_gpu_fullsize.py -x -q -k "edges or virtual_ranks or infinite or fp32_operand or na_logical or config2_f64_two" 2>&1 | tail -15
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err; echo "bench exit=$?"
NNLM_BENCH_FORCE_COMM=1 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-iters 0 > $O/bench_comm.json 2> $O/bench_comm.err; echo "bench comm exit=$?"
NNLM_BENCH_PRIME=1 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --others 0 > $O/bench_prime.json 2> $O/bench_prime.err; echo "bench prime exit=$?"
python - <<PY
import json
for f in ("bench_a", "bench_comm", "bench_prime"):
    try:
        d = json.load(open("$O/%s.json" % f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "it/s", round(d["value"], 1), "repeats", [round(x, 4) for x in d["repeats"]["ms_per_step"]], {k: round(v, 4) for k, v in d["phases_ms"].items() if k != "note"})
    for k, v in (d.get("other_configs") or {}).items():
        print("  ", k, v.get("ms_per_step"), v.get("error"))
PY
