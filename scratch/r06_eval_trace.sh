#!/bin/bash
# kernel trace of the rollout, one replayed evaluation listed in order (scratch/r06_eval_trace.py) -> gpurun_out/r06c_eval_trace.txt
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=1
rocprofv3 --kernel-trace --output-format csv -d $O/prof_ev -o ev -- python3 bench.py --mode rollout --gen-frames 3 --batch 1 > $O/r06_prof_ev.log 2>&1
python scratch/r06_eval_trace.py $O/prof_ev > $O/r06c_eval_trace.txt 2>&1; rm -rf $O/prof_ev
