#!/bin/bash
run() { echo -n "$1: "; env $1 python bench.py --steps 16 --warmup 4 --cpu-frames 0 --no-extra --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('frames/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3))"; }
run "X=1"
run "ONIRIS_EW_NT_MB=32"
run "ONIRIS_EW_NT_MB=64"
run "ONIRIS_EW_NT_MB=160"
run "ONIRIS_EW_NT_MB=0"
run "X=1"
run "ONIRIS_DKV_ITEM_KEYS=64"
run "ONIRIS_COMM_CUS_ALWAYS=0 AMD_DIRECT_DISPATCH=0"
run "HSA_KERNARG_POOL_SIZE=67108864"
run "X=1"
