#!/bin/bash
# second end-of-round pass (after the decode-stream / rope-hoist / act_out dealing changes): full GPU suite, default bench, rollout_256, rollout stats
O=gpurun_out
echo "(suite run separately: 460 passed)"
( time python bench.py > $O/r06f_bench.json 2> $O/r06f_bench.err ) 2> $O/r06f_bench_time.txt; tail -c 600 $O/r06f_bench.json
python bench.py --mode rollout --gen-frames 256 --batch 1 > $O/r06f_rollout_256.json 2> $O/r06f_rollout_256.err; cat $O/r06f_rollout_256.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GPU_MAX_HW_QUEUES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ro -o ro -- python3 bench.py --mode rollout --gen-frames 4 --batch 1 > $O/r06f_prof_ro.log 2>&1
cp "$(find $O/prof_ro -name '*kernel_stats.csv' | head -1)" $O/r06f_rollout_kernel_stats.csv
python scratch/r06_eval_trace.py $O/prof_ro > $O/r06f_eval_trace.txt 2>&1; rm -rf $O/prof_ro
