"""Neighbours of __amd_rocclr_copyBuffer dispatches in a rocprofv3 kernel trace: usage copybuf_ctx.py <kernel_trace.csv>"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"])[:50]
ctx = collections.Counter()
for i, r in enumerate(rows):
    if "copyBuffer" in r["Kernel_Name"]:
        prev = name(rows[i - 1]) if i else "-"
        nxt = name(rows[i + 1]) if i + 1 < len(rows) else "-"
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        ctx[(prev, nxt, r.get("Stream_Id", "?"), r.get("Grid_Size", "?"))] += 1
for k, v in ctx.most_common(25):
    print(v, k)
print("total copyBuffer", sum(ctx.values()), "of", len(rows), "dispatches")
