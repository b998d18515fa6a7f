#!/bin/bash
# kernel trace of the rollout, one evaluation listed in order: default tree, then with the 64-channel rounds in the few-tile 1x1 launches
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=1
rocprofv3 --kernel-trace --output-format csv -d $O/prof_ev -o ev -- python3 bench.py --mode rollout --gen-frames 3 --batch 1 > $O/r06_prof_ev.log 2>&1
python scratch/r06_eval_trace.py $O/prof_ev > $O/r06b_eval_trace.txt 2>&1; rm -rf $O/prof_ev
export ONIRIS_BIG_TILE=68
rocprofv3 --kernel-trace --output-format csv -d $O/prof_ev2 -o ev -- python3 bench.py --mode rollout --gen-frames 3 --batch 1 > $O/r06_prof_ev2.log 2>&1
python scratch/r06_eval_trace.py $O/prof_ev2 > $O/r06b_eval_trace_narrow.txt 2>&1; rm -rf $O/prof_ev2
