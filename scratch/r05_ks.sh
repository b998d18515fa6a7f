#!/bin/bash
# kernel-stats of the default bench under an environment setting:  bash scratch/r05_ks.sh TAG [ENV=VAL ...]
O=gpurun_out
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o ks -- python3 bench.py --steps 4 --warmup 2 --cpu-frames 0 --no-extra > $O/ks_$tag.log 2>&1
cp "$(find $O/prof_$tag -name '*kernel_stats.csv' | head -1)" $O/ks_$tag.csv; rm -rf $O/prof_$tag
