"""Summarise rocprofv3 --pmc csv output: per kernel name, mean counter value per dispatch.
usage: pmc_summary.py DIR [DIR ...]   (one directory per counter pass; FETCH_SIZE and WRITE_SIZE need separate passes on gfx950)

When both FETCH_SIZE and WRITE_SIZE are present a closing table gives HBM bytes per dispatch as MI355X_MICROARCH.md prescribes for
gfx950: the counters are in KiB, FETCH_SIZE tallies 128-byte requests of wide coalesced streaming reads at 64 bytes (doubled here),
WRITE_SIZE is taken as reported."""
import csv, glob, os, sys, collections

tot = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?").split("(")[0][:60]
                acc[name][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
        print("==", os.path.relpath(f, d))
        for name, cs in sorted(acc.items()):
            for c, v in cs.items():
                print(f"{name:62s} {c:12s} n={len(v):5d} mean={sum(v)/len(v):.6g} max={max(v):.6g}")
                tot[name][c] = sum(v) / len(v)
both = {n: c for n, c in tot.items() if "FETCH_SIZE" in c and "WRITE_SIZE" in c}
if both:
    print("== HBM bytes per dispatch: 2 x FETCH_SIZE + WRITE_SIZE (KiB -> MB)")
    for n, c in sorted(both.items(), key=lambda t: -(2 * t[1]["FETCH_SIZE"] + t[1]["WRITE_SIZE"])):
        mb = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e6
        if mb >= 1:
            print(f"{n:62s} {mb:10.1f} MB  (fetch x2 {2 * c['FETCH_SIZE'] * 1024 / 1e6:9.1f}, write {c['WRITE_SIZE'] * 1024 / 1e6:8.1f})")
