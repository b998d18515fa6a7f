O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ONIRIS_ONLY_MODE=2d
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_2d -o ks -- python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-extra --no-profile > $O/r04_prof_2d.log 2>&1
cp "$(find $O/prof_2d -name '*kernel_stats.csv' | head -1)" $O/r04_kernel_stats_2d.csv; rm -rf $O/prof_2d
head -32 $O/r04_kernel_stats_2d.csv | cut -c1-140
tail -2 $O/r04_prof_2d.log | cut -c1-300
