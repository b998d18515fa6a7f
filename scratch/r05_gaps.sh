#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $O/prof_gaps -o kt -- python3 bench.py --steps 8 --warmup 4 --cpu-frames 0 --no-extra --no-profile > $O/gaps.log 2>&1
f="$(find $O/prof_gaps -name '*kernel_trace.csv' | head -1)"
python3 scratch/r05_gaps.py "$f" > $O/r05_gaps.txt 2>&1
rm -rf $O/prof_gaps
tail -45 $O/r05_gaps.txt
