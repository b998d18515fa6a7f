#!/bin/bash
# second end-of-round pass (after the decode-stream / rope-hoist / act_out dealing changes): full GPU suite, default bench, rollout_256, rollout stats
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r06g_gputests.txt 2>&1; tail -3 $O/r06g_gputests.txt
( time python bench.py > $O/r06g_bench.json 2> $O/r06g_bench.err ) 2> $O/r06g_bench_time.txt; tail -c 600 $O/r06g_bench.json
python bench.py --mode rollout --gen-frames 256 --batch 1 > $O/r06g_rollout_256.json 2> $O/r06g_rollout_256.err; cat $O/r06g_rollout_256.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GPU_MAX_HW_QUEUES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ro -o ro -- python3 bench.py --mode rollout --gen-frames 4 --batch 1 > $O/r06g_prof_ro.log 2>&1
cp "$(find $O/prof_ro -name '*kernel_stats.csv' | head -1)" $O/r06g_rollout_kernel_stats.csv
python scratch/r06_eval_trace.py $O/prof_ro > $O/r06g_eval_trace.txt 2>&1; rm -rf $O/prof_ro
