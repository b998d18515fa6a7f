"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, they do not fit one pass) into
profiles/<round>_pmc_traffic.{txt,json}.  HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: the counters are
in KiB and gfx950 tallies the 128-byte read requests of wide loads at 64 B (MI355X_MICROARCH.md, HBM section).
usage: pmc_traffic.py <fetch_dir> <write_dir> <out_prefix>"""
import sys, glob, csv, collections, json

def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc

fd, wd, out = sys.argv[1:4]
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
rows = []
for k in F:
    if k in W:
        f, w = sum(F[k]) / len(F[k]), sum(W[k]) / len(W[k])
        rows.append((k, len(F[k]), f, w, (2 * f + w) * 1024, sum(F[k]) * 2 * 1024 + sum(W[k]) * 1024))
rows.sort(key=lambda r: -r[5])
with open(out + ".txt", "w") as fh:
    fh.write("rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / WRITE_SIZE (pass 2) -- python3 bench.py --steps 1 --warmup 1 "
             "--cpu-frames 0 --no-profile\nunits: KiB per launch (mean); HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
             "(gfx950: FETCH_SIZE counts 128-B requests as 64 B)\n\n")
    for k, n, f, w, b, _ in rows[:40]:
        fh.write(f"{k[:88]:88s} n={n:4d} FETCH_SIZE={f:10.0f} WRITE_SIZE={w:10.0f} -> {b/1e6:8.1f} MB/launch\n")
json.dump({k: dict(launches=n, fetch_kib=f, write_kib=w, hbm_bytes_per_launch=b) for k, n, f, w, b, _ in rows},
          open(out + ".json", "w"), indent=1)
print(open(out + ".txt").read()[:3000])
