"""Per-step table from a rocprofv3 --stats kernel_stats.csv.  usage: stats_sum.py <csv> <steps> [filter,filter...] [top]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1]))); steps = int(sys.argv[2])
flt = [f for f in (sys.argv[3].split(",") if len(sys.argv) > 3 else []) if f]
top = int(sys.argv[4]) if len(sys.argv) > 4 else 60
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel ms/step {tot/steps/1e6:.3f}")
for r in rows[:top]:
    n = re.sub(r"\(.*", "", r["Name"])[:64]
    if not flt or any(k in n for k in flt):
        print(f"{n:64s} {int(r['Calls'])/steps:6.1f} {int(r['TotalDurationNs'])/steps/1e6:7.3f} ms avg {float(r['AverageNs'])/1e3:7.1f} us")
