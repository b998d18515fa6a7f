"""Join rocprofv3 --hip-trace and --kernel-trace by correlation id: for every kernel of one steady-state step, the lead
of the host (GPU start - end of its hipLaunchKernel call).  A GPU gap in front of a kernel whose lead is ~0 is the GPU
waiting for the host.  usage: lead_trace.py <dir> [step_from_end=2]"""
import csv, glob, sys, re, collections
d = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kf = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
af = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
api = {}
for r in csv.DictReader(open(af)):
    if "Launch" in r["Function"]:
        api[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
rows = []
for r in csv.DictReader(open(kf)):
    a = api.get(r["Correlation_Id"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:50], a))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]
bursts = [ends[i] for i in range(len(ends)) if i + 1 == len(ends) or ends[i + 1] - ends[i] > 50]
a, b = bursts[-which - 1] + 1, bursts[-which] + 1
step = rows[a:b]
t0 = step[0][0]; cur = t0
print(f"step: {len(step)} launches, {(step[-1][1]-t0)/1e6:.2f} ms; matched launches: {sum(1 for r in step if r[3])}")
tot_gap = host_gap = 0
seg = collections.OrderedDict()
for i, (s, e, n, ap) in enumerate(step):
    gap = max(0, s - cur)
    lead = (s - ap[1]) if ap else None
    tot_gap += gap
    hb = gap > 3000 and lead is not None and lead < 20000
    if hb:
        host_gap += gap
    key = int((s - t0) / 1e6)          # 1 ms buckets of the step
    v = seg.setdefault(key, [0, 0, 0, []])
    v[0] += gap; v[1] += gap if hb else 0; v[2] += 1
    if lead is not None: v[3].append(lead)
    cur = max(cur, e)
print(f"idle {tot_gap/1e6:.2f} ms, of which host-bound (kernel started < 20 us after its launch returned) {host_gap/1e6:.2f} ms")
print(" ms   launches  idle_us  hostbound_us  median_lead_us")
for k, v in seg.items():
    ld = sorted(v[3]); med = ld[len(ld)//2] / 1e3 if ld else -1
    print(f"{k:4d}   {v[2]:5d}   {v[0]/1e3:8.1f}   {v[1]/1e3:8.1f}   {med:10.1f}")
