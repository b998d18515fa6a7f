"""Where a generated frame's time goes outside its 31 evaluations: from a rocprofv3 --kernel-trace of `bench.py --mode rollout`,
per frame: span, busy time, the idle gaps above 20 us and the kernels that run between two evaluations' groups.
usage: frame_gaps.py DIR"""
import csv, glob, sys, re, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")[:60]
ev = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("dart_input_kernel")]
# evaluations come in groups of 31 (one frame); a frame boundary = more than 141 kernels between two dart_input launches, or a long gap
frames, cur = [], [ev[0]]
for a, b in zip(ev, ev[1:]):
    if b - a > 160 or S(rows[b]) - E(rows[b - 1]) > 200_000:
        frames.append(cur); cur = []
    cur.append(b)
frames.append(cur)
for k, fr in enumerate(frames[-3:-1]):
    nxt = frames[len(frames) - 3 + k + 1][0]
    a, b = fr[0], nxt
    seg = rows[a:b]
    span = S(rows[b]) - S(rows[a]); busy = sum(E(r) - S(r) for r in seg)
    print(f"frame: {len(fr)} evaluations, {len(seg)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms")
    last_eval_end = fr[-1] + 141
    tail = rows[last_eval_end:b]
    c = collections.Counter(short(r["Kernel_Name"]) for r in tail)
    tb = sum(E(r) - S(r) for r in tail)
    print(f"  behind the last evaluation: {len(tail)} kernels, busy {tb/1e3:.0f} us, span {(S(rows[b]) - E(rows[last_eval_end - 1]))/1e3:.0f} us")
    for n, v in c.most_common(8):
        print(f"     {v:4d} x {n}")
    gaps = sorted(((S(y) - E(x), short(x["Kernel_Name"]), short(y["Kernel_Name"])) for x, y in zip(seg, seg[1:])), reverse=True)[:6]
    for g, x, y in gaps:
        print(f"  gap {g/1e3:7.1f} us between {x} and {y}")
