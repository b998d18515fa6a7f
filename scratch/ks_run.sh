#!/bin/bash
# usage: ks_run.sh TAG  -> gpurun_out/ks_TAG.csv (rocprofv3 kernel stats of a short bench) + bench line
O=gpurun_out; T=$1
python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-profile > $O/bench_$T.json 2> $O/bench_$T.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o ks -- python3 bench.py --steps 8 --warmup 4 --cpu-frames 0 --no-profile > $O/prof_$T.log 2>&1
cp "$(find $O/prof_$T -name '*kernel_stats.csv' | head -1)" $O/ks_$T.csv; rm -rf $O/prof_$T
python - <<PY
import json
d=json.load(open("$O/bench_$T.json")); print(d["value"], d["ms_per_step"], d["config"]["ms_3d_step"], d["config"]["ms_2d_step"])
PY
