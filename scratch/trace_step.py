"""Timeline of ONE training step out of a rocprofv3 kernel trace: every idle gap > thr us with its position in the step.
usage: trace_step.py <dir> [which_step_from_end=2] [thr_us=5]"""
import csv, glob, sys, re
d = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 2; thr = float(sys.argv[3]) if len(sys.argv) > 3 else 5
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:60]) for r in csv.DictReader(open(f)))
ends = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]
# steps end at the last adamw launch of a burst
bursts = [ends[i] for i in range(len(ends)) if i + 1 == len(ends) or ends[i + 1] - ends[i] > 50]
a, b = bursts[-which - 1] + 1, bursts[-which] + 1
step = rows[a:b]
t0 = step[0][0]; cur = t0; idle = 0; busy = 0
print(f"step: {len(step)} launches, {(step[-1][1]-t0)/1e6:.2f} ms")
for i, (s, e, n) in enumerate(step):
    if s - cur > thr * 1e3:
        idle += s - cur
        print(f"  t={(cur-t0)/1e6:7.3f} ms  gap {(s-cur)/1e3:6.1f} us   after #{i-1} {step[i-1][2] if i else '-':40s} -> before {n}")
    cur = max(cur, e)
print(f"idle in gaps > {thr} us: {idle/1e6:.2f} ms")
