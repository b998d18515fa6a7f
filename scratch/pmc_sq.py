"""Summarise rocprofv3 --pmc passes of SQ counters per kernel: mean per launch of every counter found under the given directories.
usage: pmc_sq.py <out.txt> <dir> [<dir> ...] [--only substring]"""
import sys, glob, csv, collections
args = sys.argv[1:]
only = None
if "--only" in args:
    i = args.index("--only"); only = args[i + 1]; del args[i:i + 2]
out, dirs = args[0], args[1:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if only and only not in r["Kernel_Name"]:
                continue
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out, "w") as fh:
    for k in sorted(acc):
        fh.write(k[:120] + "\n")
        for c in sorted(acc[k]):
            v = acc[k][c]
            fh.write(f"    {c:36s} {sum(v) / len(v):16.0f}  (n={len(v)})\n")
print(open(out).read()[:6000])
