#!/bin/bash
# rollout frames/s under runtime knobs that relax the launch-path throttle (profiles/r02_host_side.txt item 3)
run() { echo -n "$1: "; env $1 python bench.py --mode rollout --gen-frames 48 --batch 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'frames/s', round(d['ms_per_unet_eval'],4),'ms/eval')"; }
run "X=1"
run "HSA_KERNARG_POOL_SIZE=67108864"
run "HSA_KERNARG_POOL_SIZE=268435456"
run "ROC_SIGNAL_POOL_SIZE=4096"
run "GPU_MAX_COMMAND_BUFFERS=64"
run "ROC_AQL_QUEUE_SIZE=65536"
run "DEBUG_HIP_GRAPH_BATCH_SIZE=0"
run "AMD_DIRECT_DISPATCH=0"
run "X=1"
