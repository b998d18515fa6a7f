#!/bin/bash
# dK/dV item size A/B (ONIRIS_DKV_ITEM_KEYS=64/128): parity tests under both, then kernel durations of scratch/attn_bench.py at B = 2 and 8
O=gpurun_out
for k in 64 128; do
ONIRIS_DKV_ITEM_KEYS=$k python -m pytest tests/test_ops_gpu.py tests/test_verification_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "attention or attn or g6" 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 2 8; do for v in 64 128; do
  export ONIRIS_DKV_ITEM_KEYS=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dkv$v -o ks -- python3 scratch/attn_bench.py 12 $B > $O/dkv$v.log 2>&1
  python3 - $O/prof_dkv$v $v $B <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attn_" in r["Name"]:
        print(f"B={sys.argv[3]} keys={sys.argv[2]:>3} {r['Name'][:58]:58s} calls {r['Calls']:>3} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
  rm -rf $O/prof_dkv$v
done; done
