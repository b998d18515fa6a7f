#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=1
rocprofv3 --kernel-trace --output-format csv -d $O/prof_g -o g -- python3 bench.py --mode rollout --gen-frames 6 --batch 1 > $O/r06_prof_g.log 2>&1
python scratch/r06_rollout_gaps.py $O/prof_g > $O/r06c_rollout_gaps.txt 2>&1; rm -rf $O/prof_g
