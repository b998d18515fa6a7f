# same-box A/B: round-3 kernels (tile kernels everywhere) vs the streaming kernels of the 32-channel level
for b in 8 2; do
for mode in new old; do
  if [ $mode = old ]; then export ONIRIS_BIG_TILE=3 ONIRIS_WGRAD=2; else unset ONIRIS_BIG_TILE ONIRIS_WGRAD; fi
  python bench.py --steps 12 --warmup 4 --cpu-frames 0 --no-extra --no-profile --batch $b > gpurun_out/r04_ab_${mode}_b$b.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/r04_ab_${mode}_b$b.json'));print('$mode', $b, round(d['value']), round(d['ms_per_step'],2), d['config']['ms_3d_step'], d['config']['ms_2d_step'])"
done; done
