"""Where a generated frame's wall time goes in the rollout (rocprofv3 --kernel-trace csv of bench.py --mode rollout): per frame (31 evaluations =
31 dart_input_kernel launches) the span, the kernel-busy time, the kernels outside the evaluation graphs and the idle gaps > 20 us.
usage: r06_rollout_gaps.py DIR"""
import csv, glob, sys, re, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
S = [int(r["Start_Timestamp"]) for r in rows]; E = [int(r["End_Timestamp"]) for r in rows]
N = [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60] for r in rows]
ev = [i for i, n in enumerate(N) if n.startswith("dart_input_kernel")]
# frames: groups of 31 evaluations, counted from the end (the prefill / warm-up frames come first)
nfr = len(ev) // 31
for fi in range(max(0, nfr - 3), nfr):
    a = ev[len(ev) - (nfr - fi) * 31]
    b = ev[len(ev) - (nfr - fi - 1) * 31] if fi + 1 < nfr else len(rows) - 1
    span = (S[b] - S[a]) / 1e3
    busy = sum(E[i] - S[i] for i in range(a, b)) / 1e3
    gaps = [(S[i + 1] - E[i]) / 1e3 for i in range(a, b)]
    big = [(g, N[i], N[i + 1]) for i, g in zip(range(a, b), gaps) if g > 20]
    print(f"frame {fi}: {b - a} kernels, span {span:.0f} us, busy {busy:.0f} us, idle {span - busy:.0f} us; gaps > 20 us: {len(big)}, sum {sum(g for g, _, _ in big):.0f} us")
    c = collections.Counter()
    for g, x, y in big:
        c[(x, y)] += g
    for (x, y), g in c.most_common(6):
        print(f"      {g:8.0f} us between {x}  ->  {y}")
    # kernels of the frame that are not part of the 31 evaluation graphs: everything after the 31st precond_out until the next dart_input
    last = max(i for i in range(a, b) if N[i].startswith("_Z18precond_out") or N[i].startswith("precond_out"))
    tail = collections.Counter()
    for i in range(last + 1, b):
        tail[N[i]] += (E[i] - S[i]) / 1e3
    print(f"      after the last evaluation: {b - last - 1} kernels, {sum(tail.values()):.0f} us busy, span {(S[b] - E[last]) / 1e3:.0f} us")
    for n, t in tail.most_common(6):
        print(f"          {t:7.0f} us  {n}")
