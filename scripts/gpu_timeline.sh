#!/bin/bash
# Kernel timeline of a few dense iterations (rocprofv3 --kernel-trace): start, duration and gap to the previous kernel's end.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tl
cd /tmp
ITERS=3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o tl -- python $R/scripts/${TL_SCRIPT:-gpu_time.py} > $R/gpurun_out/tl/run.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last ~40 kernels = the final iterations
tail = rows[-int(__import__("os").environ.get("TL_ROWS", "46")):]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  q{r.get('Queue_Id','?'):>2}  {name}")
    prev_end = max(prev_end or 0, e)
PY
rm -f gpurun_out/tl/*kernel_trace.csv
