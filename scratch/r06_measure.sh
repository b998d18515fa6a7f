#!/bin/bash
# end-of-round measurements (run through gpurun from the repo root); everything lands in gpurun_out/
set -x
O=gpurun_out
( time python bench.py --steps 20 --warmup 5 > $O/r06_bench.json 2> $O/r06_bench.err ) 2> $O/r06_bench_time.txt
python bench.py --batch 2 --steps 20 --warmup 5 --cpu-frames 0 --no-extra > $O/r06_bench_b2.json 2> $O/r06_bench_b2.err
python bench.py --net cs --steps 8 --warmup 4 --no-profile > $O/r06_bench_cs.json 2> $O/r06_bench_cs.err
python bench.py --net cs --frames 64 --steps 8 --warmup 4 --no-profile > $O/r06_bench_cs_t64.json 2> $O/r06_bench_cs_t64.err
python bench.py --mode rollout --gen-frames 256 --batch 1 > $O/r06_rollout_256.json 2> $O/r06_rollout_256.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ks -o ks -- python3 bench.py --steps 4 --warmup 2 --cpu-frames 0 --no-extra > $O/r06_prof_ks.log 2>&1
cp "$(find $O/prof_ks -name '*kernel_stats.csv' | head -1)" $O/r06_kernel_stats.csv; rm -rf $O/prof_ks
ONIRIS_ONLY_MODE=2d rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ks2 -o ks -- python3 bench.py --steps 8 --warmup 2 --cpu-frames 0 --no-extra --no-profile > $O/r06_prof_ks2.log 2>&1
cp "$(find $O/prof_ks2 -name '*kernel_stats.csv' | head -1)" $O/r06_kernel_stats_2d.csv; rm -rf $O/prof_ks2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_f -o f -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-profile --no-extra > $O/r06_prof_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_w -o w -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-profile --no-extra > $O/r06_prof_w.log 2>&1
python scratch/pmc_traffic.py $O/prof_f $O/prof_w $O/r06_pmc_traffic > /dev/null 2>&1
rm -rf $O/prof_f $O/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ro -o ro -- python3 bench.py --mode rollout --gen-frames 4 --batch 1 > $O/r06_prof_ro.log 2>&1
cp "$(find $O/prof_ro -name '*kernel_stats.csv' | head -1)" $O/r06_rollout_kernel_stats.csv; rm -rf $O/prof_ro
ONIRIS_PROFILE_SHAPES=1 python bench.py --steps 4 --warmup 2 --cpu-frames 0 --no-extra > /dev/null 2> $O/r06_conv_shapes.txt
python scratch/r06_frame_attn_ab.py > $O/r06_ab_frame_attn.txt 2>&1
