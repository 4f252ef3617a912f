#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
for m in 1 0; do
(cd /tmp && NNLM_EXP_PLAN2_OFF=$m timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05/prof_q$m -o p -- python $R/bench.py --config 2 --steps 20 --warmup 2 --cpu-iters 0 --repeats 1 --others 0 --call 0 > /dev/null 2> /dev/null)
f=$(find $R/gpurun_out/r05/prof_q$m -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
d = collections.defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"]
    if "xprod16" in nm or "sweep_scd" in nm:
        key = (nm.split("(")[0][:40], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"))
        d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("off=$m")
for k, v in sorted(d.items()):
    v.sort()
    print("  ", k, len(v), "median", v[len(v)//2] / 1e3, "min", v[0] / 1e3)
PY
rm -rf $R/gpurun_out/r05/prof_q$m
done
