#!/bin/bash
for i in 1 2; do for e in "X=1" "GPU_MAX_HW_QUEUES=1"; do
  echo -n "$e (run $i): "; env $e python bench.py --steps 16 --warmup 4 --cpu-frames 0 --no-extra --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('frames/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3))"
done; done
