#!/bin/bash
# host-side runtime knobs, same box: value, ms/step, host enqueue
run() { echo "== $*"; "$@" 2>&1 >/tmp/o.json | grep "host enq"; python -c "import json; d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('   ', round(d['value'],1), round(d['ms_per_step'],3))"; }
export ONIRIS_HOST_TIMING=1
A="bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-profile"
run python $A
run taskset -c 0-7 python $A
run taskset -c 2 python $A
run env AMD_DIRECT_DISPATCH=0 python $A
run env ROC_CPU_WAIT_FOR_SIGNAL=0 python $A
run env ROC_CPU_WAIT_FOR_SIGNAL=1 python $A
run env HSA_ENABLE_INTERRUPT=0 python $A
run env GPU_MAX_HW_QUEUES=1 python $A
run python $A
