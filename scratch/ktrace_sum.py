"""Mean kernel duration per (kernel name, grid) from a rocprofv3 --kernel-trace csv dir.  usage: ktrace_sum.py <dir> [filter]"""
import sys, glob, csv, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:70], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g), v in sorted(acc.items()):
    v = sorted(v)
    print(f"{k:72s} grid={g:>9s} n={len(v):3d} median={v[len(v)//2]/1e3:8.1f} us min={v[0]/1e3:8.1f}")
