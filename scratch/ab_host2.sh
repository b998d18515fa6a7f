#!/bin/bash
# launch-throttle knobs of the HIP runtime, same box: value, ms/step, host enqueue
run() { echo "== $*"; env "$@" ONIRIS_HOST_TIMING=1 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-profile 2>&1 >/tmp/o.json | grep "host enq"; python -c "import json; d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('   ', round(d['value'],1), round(d['ms_per_step'],3))"; }
run X=1
run DEBUG_CLR_MAX_BATCH_SIZE=100000
run DEBUG_CLR_MAX_BATCH_SIZE=10
run DEBUG_CLR_BATCH_CPU_SYNC_SIZE=1000000
run DEBUG_CLR_BATCH_CPU_SYNC_SIZE=16
run HSA_KERNARG_POOL_SIZE=67108864
run HSA_KERNARG_POOL_SIZE=65536
run ROC_SIGNAL_POOL_SIZE=8192
run ROC_SIGNAL_POOL_SIZE=16
run GPU_MAX_COMMAND_BUFFERS=64
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run ROC_USE_FGS_KERNARG=0
run ROC_SKIP_KERNEL_ARG_COPY=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=0
run X=1
