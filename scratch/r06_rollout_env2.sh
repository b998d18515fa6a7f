#!/bin/bash
# rollout frames/s under the runtime's fence / kernarg / graph-packet knobs (the per-node floor of a replayed evaluation is ~4.6 us)
run() { echo -n "$1: "; timeout 180 env $1 python bench.py --mode rollout --gen-frames 24 --batch 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'frames/s', round(d['ms_per_unet_eval'],4),'ms/eval')" || echo failed; }
run "X=1"
run "AMD_OPT_FLUSH=0"
run "AMD_OPT_FLUSH=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "ROC_USE_FGS_KERNARG=0"
run "ROC_SYSTEM_SCOPE_SIGNAL=0"
run "DEBUG_HIP_KERNARG_COPY_OPT=0"
run "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0"
run "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1"
run "ROC_ACTIVE_WAIT_TIMEOUT=1000"
run "X=1"
