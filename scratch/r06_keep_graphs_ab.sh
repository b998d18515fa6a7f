#!/bin/bash
# rollout with the frames' graphs destroyed as soon as finished (default) vs never (ONIRIS_SAMPLER_KEEP_GRAPHS=1), 8 and 64 frames, + reserved memory
run() { echo -n "$1 frames=$2: "; env $1 python bench.py --mode rollout --gen-frames $2 --batch 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'frames/s', round(d['ms_per_unet_eval'],4),'ms/eval')"; }
run "ONIRIS_SAMPLER_KEEP_GRAPHS=0" 8
run "ONIRIS_SAMPLER_KEEP_GRAPHS=1" 8
run "ONIRIS_SAMPLER_KEEP_GRAPHS=0" 64
run "ONIRIS_SAMPLER_KEEP_GRAPHS=1" 64
run "ONIRIS_SAMPLER_KEEP_GRAPHS=1 ROC_AQL_QUEUE_SIZE=65536" 64
