"""HIP API trace (rocprofv3 --hip-trace): what the host was doing in the tail window -- slow calls, synchronous copies."""
import csv, glob, sys, collections
d, win = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
rows.sort()
t_end = rows[-1][1]
# the process tail has teardown; anchor the window at the last hipLaunchKernel
t_end = max(r[1] for r in rows if r[2] == "hipLaunchKernel")
rows = [r for r in rows if r[0] >= t_end - win * 1e6 and r[1] <= t_end]
per = collections.defaultdict(list)
for s, e, n in rows:
    per[n].append(e - s)
print(f"window {win} ms: {len(rows)} API calls")
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v.sort()
    print(f"{n:32s} n={len(v):6d} total={sum(v)/1e6:8.2f} ms  median={v[len(v)//2]/1e3:7.1f} us  p99={v[int(len(v)*0.99)]/1e3:8.1f} us  max={v[-1]/1e3:9.1f} us")
slow = [(s, e, n) for s, e, n in rows if e - s > 30000]
print("calls > 30 us:", len(slow), "total", sum(e - s for s, e, n in slow) / 1e6, "ms")
t0 = rows[0][0]
for s, e, n in slow[:60]:
    print(f"   t={(s-t0)/1e6:8.3f} ms  {(e-s)/1e3:9.1f} us  {n}")
