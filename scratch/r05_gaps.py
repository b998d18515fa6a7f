"""Idle time of the GPU between the kernels of a training step, from a rocprofv3 --kernel-trace CSV.
python scratch/r05_gaps.py trace.csv > report   (steps are delimited by weight_bwd_kernel launches)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
# steps: from one adamw/weight_bwd to the next
cut = [i for i, k in enumerate(ks) if k[2].startswith("weight_bwd_kernel")]
print("kernels", len(ks), "steps", len(cut))
def short(n):
    return n.replace("void ", "").split("(")[0][:48]
for a, b in zip(cut[:-1], cut[1:]):          # every step (warm-up 4, timed 8, then bench.py's six single synchronised steps)
    seg = ks[a:b]
    wall = seg[-1][1] - seg[0][0]
    busy = 0; end = seg[0][0]; gaps = []
    for s, e, n in seg:
        if s > end:
            gaps.append((s - end, n))
        busy += max(0, e - max(s, end)); end = max(end, e)
    idle = wall - busy
    big = sorted(gaps, reverse=True)[:6]
    hist = collections.Counter()
    for g, _ in gaps:
        hist["<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"] += g
    print(f"step: {len(seg)} launches wall {wall/1e6:.2f} ms busy {busy/1e6:.2f} idle {idle/1e6:.2f} ms ({100*idle/wall:.1f} %)  by gap size (ms):",
          {k: round(v / 1e6, 2) for k, v in sorted(hist.items())}, " largest:", [(round(g / 1e3), short(n)) for g, n in big])
# which kernels follow the idle time (summed over the last 8 steps)
a, b = cut[4], cut[12]          # the timed steps
seg = ks[a:b]; end = seg[0][0]; by = collections.Counter(); cnt = collections.Counter(); prev = None; byprev = collections.Counter()
for s, e, n in seg:
    if s > end:
        by[short(n)] += s - end; cnt[short(n)] += 1
        if prev: byprev[short(prev)] += s - end
    if e > end: prev = n
    end = max(end, e)
print("idle time in FRONT of (ms per step, gaps per step):")
for n, v in by.most_common(25):
    print(f"  {n:50s} {v/8e6:7.3f} {cnt[n]/8:7.1f}  avg {v/cnt[n]/1e3:6.1f} us")
print("idle time BEHIND:")
for n, v in byprev.most_common(12):
    print(f"  {n:50s} {v/8e6:7.3f}")
