#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cs -o ks -- python3 bench.py --net cs --steps 4 --warmup 2 --cpu-frames 0 --no-profile > $O/prof_cs.log 2>&1
cp "$(find $O/prof_cs -name '*kernel_stats.csv' | head -1)" $O/ks_cs.csv; rm -rf $O/prof_cs
tail -1 $O/prof_cs.log | cut -c1-400
