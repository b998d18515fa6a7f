#!/bin/bash
# end-of-round measurements (run through gpurun from the repo root); everything lands in gpurun_out/
set -x
O=gpurun_out
python -m pytest tests -m gpu -q > $O/r03_gputests_final.log 2>&1; echo rc=$? >> $O/r03_gputests_final.log
python bench.py --steps 20 --warmup 5 > $O/r03_bench.json 2> $O/r03_bench.err
python bench.py --net cs --steps 8 --warmup 4 --no-profile > $O/r03_bench_cs.json 2> $O/r03_bench_cs.err
python bench.py --steps 16 --warmup 8 --accum 4 --cpu-frames 0 --no-profile > $O/r03_bench_accum4.json 2> $O/r03_bench_accum4.err
python bench.py --mode rollout --gen-frames 16 --batch 1 > $O/r03_rollout.json 2> $O/r03_rollout.err
python bench.py --mode rollout --gen-frames 256 --batch 1 > $O/r03_rollout_256.json 2> $O/r03_rollout_256.err
ONIRIS_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 4 --warmup 2 --cpu-frames 0 --no-profile > $O/r03_selflaunch2.json 2> $O/r03_selflaunch2.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ks -o ks -- python3 bench.py --steps 4 --warmup 2 --cpu-frames 0 > $O/r03_prof_ks.log 2>&1
cp "$(find $O/prof_ks -name '*kernel_stats.csv' | head -1)" $O/r03_kernel_stats.csv; rm -rf $O/prof_ks
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_f -o f -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-profile > $O/r03_prof_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_w -o w -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-profile > $O/r03_prof_w.log 2>&1
python scratch/pmc_traffic.py $O/prof_f $O/prof_w $O/r03_pmc_traffic > /dev/null 2>&1
rm -rf $O/prof_f $O/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ro -o ro -- python3 bench.py --mode rollout --gen-frames 4 --batch 1 > $O/r03_prof_ro.log 2>&1
cp "$(find $O/prof_ro -name '*kernel_stats.csv' | head -1)" $O/r03_rollout_kernel_stats.csv; rm -rf $O/prof_ro
tail -3 $O/r03_gputests_final.log
