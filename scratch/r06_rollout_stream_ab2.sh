#!/bin/bash
run() { echo -n "$1: "; env $1 python bench.py --mode rollout --gen-frames 48 --batch 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'frames/s', round(d['ms_per_unet_eval'],4),'ms/eval', d['finite'])"; }
run "GPU_MAX_HW_QUEUES=1"
run "GPU_MAX_HW_QUEUES=1 ROC_AQL_QUEUE_SIZE=65536"
run "GPU_MAX_HW_QUEUES=1 ONIRIS_SAMPLER_KEEP_GRAPHS=1"
run "GPU_MAX_HW_QUEUES=1 ONIRIS_SAMPLER_KEEP_GRAPHS=1 ROC_AQL_QUEUE_SIZE=65536"
run "GPU_MAX_HW_QUEUES=2 ONIRIS_SAMPLER_KEEP_GRAPHS=1 ROC_AQL_QUEUE_SIZE=65536"
run "GPU_MAX_HW_QUEUES=1 ONIRIS_SAMPLER_KEEP_GRAPHS=1 ROC_AQL_QUEUE_SIZE=65536 AMD_DIRECT_DISPATCH=0"
run "GPU_MAX_HW_QUEUES=1 ONIRIS_SAMPLER_KEEP_GRAPHS=1 ROC_AQL_QUEUE_SIZE=16384"
