"""Per-category busy time and idle gaps over the tail window of a rocprofv3 kernel trace.
usage: trace_gaps.py <dir with *_kernel_trace.csv> <window_ms> [n_steps]"""
import csv, glob, sys, collections, re
d, win = sys.argv[1], float(sys.argv[2]); nst = int(sys.argv[3]) if len(sys.argv) > 3 else 1
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_end = max(r[1] for r in rows); t0 = t_end - int(win * 1e6)
rows = [r for r in rows if r[0] >= t0]
busy = 0; cur_e = rows[0][0]; gaps = []
per = collections.Counter(); cnt = collections.Counter()
for s, e, n in rows:
    n = re.sub(r"\(.*", "", n)[:80]
    per[n] += e - s; cnt[n] += 1
    if s > cur_e: gaps.append(s - cur_e)
    if e > cur_e: busy += e - max(s, cur_e); cur_e = e
wall = rows[-1][1] - rows[0][0]
print(f"window {wall/1e6:.2f} ms, {len(rows)} launches, busy {busy/1e6:.2f} ms ({100*busy/wall:.1f} %), idle {sum(gaps)/1e6:.2f} ms in {len(gaps)} gaps "
      f"(>5us: {sum(g for g in gaps if g>5000)/1e6:.2f} ms, n={sum(1 for g in gaps if g>5000)})")
tot = sum(per.values())
print(f"sum of kernel durations {tot/1e6:.2f} ms; per step {tot/1e6/nst:.2f} ms")
for n, t in per.most_common(45):
    print(f"{t/1e6/nst:8.3f} ms/step {100*t/tot:5.1f} %  n/step={cnt[n]/nst:6.1f}  avg={t/cnt[n]/1e3:7.1f} us  {n}")
# which launches FOLLOW the idle gaps (the launch the host was late with)
after = collections.Counter(); aftert = collections.Counter(); cur_e = rows[0][0]
for s, e, n in rows:
    n = re.sub(r"\(.*", "", n)[:80]
    if s - cur_e > 5000: after[n] += 1; aftert[n] += s - cur_e
    cur_e = max(cur_e, e)
print("idle gaps > 5 us, by the kernel that follows:")
for n, t in aftert.most_common(25):
    print(f"{t/1e6/nst:8.3f} ms/step n/step={after[n]/nst:5.1f} avg={t/after[n]/1e3:6.1f} us  {n}")
