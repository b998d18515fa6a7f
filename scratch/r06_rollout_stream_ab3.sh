#!/bin/bash
run() { echo -n "$1: "; env $1 python bench.py --mode rollout --gen-frames 48 --batch 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'frames/s', round(d['ms_per_unet_eval'],4),'ms/eval', d['finite'])"; }
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=1"
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=0"
run "GPU_MAX_HW_QUEUES=1 ONIRIS_SAMPLER_SAME_STREAM=0"
run "GPU_MAX_HW_QUEUES=1"
echo -n "B=8 default: "; python bench.py --mode rollout --gen-frames 8 --batch 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2))"
echo -n "B=8 GPU_MAX_HW_QUEUES=1: "; GPU_MAX_HW_QUEUES=1 python bench.py --mode rollout --gen-frames 8 --batch 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2))"
