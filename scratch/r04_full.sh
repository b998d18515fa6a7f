O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r04_t2.log 2>&1; tail -4 $O/r04_t2.log
python bench.py --steps 8 --warmup 4 --cpu-frames 0 > $O/r04_b2.json 2> $O/r04_b2.err; tail -3 $O/r04_b2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_b2.json'))
print(round(d['value']), round(d['ms_per_step'],2), d['config']['ms_3d_step'], d['config']['ms_2d_step'], 'step frac', round(d['roofline_step']['frac'],3))
for k,v in d['kernels'].items(): print("  ",k, v)
print(d['extra'])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ks -o ks -- python3 bench.py --steps 4 --warmup 2 --cpu-frames 0 --no-extra --no-profile > $O/r04_prof_ks.log 2>&1
cp "$(find $O/prof_ks -name '*kernel_stats.csv' | head -1)" $O/r04_kernel_stats_b8_mid.csv; rm -rf $O/prof_ks
head -40 $O/r04_kernel_stats_b8_mid.csv | cut -c1-150
