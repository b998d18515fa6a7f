cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
export ONIRIS_CAT_ACT_FUSED=$v
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ro$v -o ro -- python3 bench.py --mode rollout --gen-frames 4 --batch 1 > gpurun_out/ro_prof$v.log 2>&1
python3 - gpurun_out/prof_ro$v $v <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows); n = sum(int(r["Calls"]) for r in rows)
print("fused", sys.argv[2], "launches", n, "kernel ms", round(tot / 1e6, 2))
for r in rows[:9]:
    print("   ", r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Name"][:70])
PY
rm -rf gpurun_out/prof_ro$v
done
