"""Ordered kernel list of the last steps of a rocprofv3 --kernel-trace run of bench.py (csv).  usage: trace_extract.py DIR OUT"""
import csv, glob, sys, re
d, out = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("dart_input_kernel")]
def short(n):
    n = re.sub(r"\(.*", "", n); n = n.replace("void ", "")
    n = re.sub(r"at::native::", "", n)
    return n[:70]
with open(out, "w") as fo:
    for a, b in zip(idx[-5:-1], idx[-4:]):
        seg = rows[a:b]
        t0 = int(seg[0]["Start_Timestamp"])
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
        fo.write(f"=== step: {len(seg)} kernels, span {(int(rows[b]['Start_Timestamp'])-t0)/1e3:.0f} us, busy {busy/1e3:.0f} us\n")
        prev_end = t0
        for r in seg:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            fo.write(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} gap{(s-prev_end)/1e3:6.1f} g{r['Grid_Size_X']:>7s} {short(r['Kernel_Name'])}\n")
            prev_end = e
