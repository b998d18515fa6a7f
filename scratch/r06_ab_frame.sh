#!/bin/bash
# same-box A/B of the FrameAttention kernels (csrc/attention_frame.h): whole training bench, default batch
for i in 1 2; do
  for v in 0 1; do
    echo "== ONIRIS_FRAME_KERNEL=$v (run $i)"
    ONIRIS_FRAME_KERNEL=$v python bench.py --steps 16 --warmup 4 --cpu-frames 0 --no-extra --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  frames/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'ms_3d', d.get('ms_3d_step'), 'ms_2d', d.get('ms_2d_step'))"
  done
done
