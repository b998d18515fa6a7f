O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cs -o ks -- python3 bench.py --net cs --steps 6 --warmup 2 --cpu-frames 0 --no-extra --no-profile > $O/r04_prof_cs.log 2>&1
cp "$(find $O/prof_cs -name '*kernel_stats.csv' | head -1)" $O/r04_kernel_stats_cs.csv; rm -rf $O/prof_cs
head -28 $O/r04_kernel_stats_cs.csv | cut -c1-130
tail -1 $O/r04_prof_cs.log | cut -c1-400
