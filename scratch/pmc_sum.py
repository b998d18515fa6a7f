"""Summarise rocprofv3 --pmc output: mean counter value per kernel name.  usage: pmc_sum.py <dir> [name-filter]"""
import sys, glob, csv, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt in k:
            acc[(k, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), cs in sorted(acc.items()):
    print(f"{k[:100]} grid={g}")
    for c, v in sorted(cs.items()):
        print(f"    {c:36s} {sum(v)/len(v):16.0f}  (n={len(v)})")
