#!/bin/bash
# usage: ks_ab.sh ENVVAR VAL1 VAL2 PATTERN  -> average kernel time of kernels matching PATTERN under each setting
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $2 $3; do
  export $1=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ab -o ks -- python3 bench.py --steps 8 --warmup 4 --cpu-frames 0 --no-profile > $O/prof_ab.log 2>&1
  f="$(find $O/prof_ab -name '*kernel_stats.csv' | head -1)"
  echo "== $1=$v"; python3 - "$f" "$4" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%-60s n=%5s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/prof_ab
done
