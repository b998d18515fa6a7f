"""Ordered kernel list of ONE replayed UNet evaluation of the rollout (rocprofv3 --kernel-trace csv of bench.py --mode rollout) + per-kernel sums.
usage: r06_eval_trace.py DIR [which_from_end]"""
import csv, glob, sys, re, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
S = [int(r["Start_Timestamp"]) for r in rows]; E = [int(r["End_Timestamp"]) for r in rows]
N = [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:70] for r in rows]
G = [(int(r.get("Grid_Size_X", 0) or 0), int(r.get("Grid_Size_Y", 0) or 0), int(r.get("Grid_Size_Z", 0) or 0)) for r in rows]
ev = [i for i, n in enumerate(N) if n.startswith("dart_input_kernel")]
a, b = ev[-k - 1], ev[-k]
print(f"=== evaluation: {b - a} kernels, span {(S[b] - S[a]) / 1e3:.0f} us, busy {sum(E[i] - S[i] for i in range(a, b)) / 1e3:.0f} us")
for i in range(a, b):
    gap = (S[i] - E[i - 1]) / 1e3 if i > a else 0.0
    print(f"{(S[i] - S[a]) / 1e3:9.1f} {(E[i] - S[i]) / 1e3:7.1f} gap {gap:5.1f} g {G[i][0]:6d}x{G[i][1]}x{G[i][2]} {N[i]}")
tot, cnt = collections.Counter(), collections.Counter()
for i in range(a, b):
    tot[N[i]] += (E[i] - S[i]) / 1e3; cnt[N[i]] += 1
print("--- sums")
for n, t in tot.most_common():
    print(f"{t:8.1f} {cnt[n]:4d} {t / cnt[n]:6.2f}  {n}")
