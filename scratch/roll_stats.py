"""Summarise a rocprofv3 --kernel-trace --stats run of `bench.py --mode rollout --gen-frames G` (per UNet evaluation)."""
import csv, sys
path, frames = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(int(r["TotalDurationNs"]) for r in rows); n = sum(int(r["Calls"]) for r in rows)
ev = (frames + 2) * 31
print(f"kernel ms total {tot/1e6:.1f}, launches {n}; per UNet eval ({ev} evals + prefill): {tot/1e6/ev:.3f} ms, {n/ev:.0f} launches")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls'])/ev:6.1f}/eval {int(r['TotalDurationNs'])/1e6/ev*1e3:8.1f} us/eval avg {float(r['AverageNs'])/1e3:7.1f} us")
