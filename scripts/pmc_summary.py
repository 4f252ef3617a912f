"""Summarise rocprofv3 --pmc csv output: per kernel name, mean counter value per dispatch."""
import csv, glob, os, sys, collections
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?").split("(")[0][:60]
            acc[name][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
    print("==", os.path.basename(f))
    for name, cs in sorted(acc.items()):
        for c, v in cs.items():
            print(f"{name:62s} {c:12s} n={len(v):5d} mean={sum(v)/len(v):.6g} max={max(v):.6g}")
