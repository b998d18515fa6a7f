#!/bin/bash
# graph-replay knobs of the HIP runtime, same box
run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-profile --graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('ms_3d_step'), d['config'].get('ms_2d_step'), round(d['loss'],4))"; }
run X=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_HIP_GRAPH_BATCH_SIZE=4096
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run ROC_AQL_QUEUE_SIZE=65536
