#!/bin/bash
# plain streaming kernel on (default) / off (ONIRIS_BIG_TILE=132: bit 7 sends those launches back to the tile kernel)
O=gpurun_out
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "test_conv_plain or test_conv_epilogues or clip_flags" 2>&1 | tail -15
for rep in 1 2; do
for bt in 132 4; do
  ONIRIS_BIG_TILE=$bt python bench.py --steps 12 --warmup 4 --cpu-frames 0 --no-extra > $O/ab5_${bt}_$rep.json 2>/dev/null
  python - $bt $rep <<'PY'
import json, sys
bt, rep = sys.argv[1:]
d = json.load(open(f"gpurun_out/ab5_{bt}_{rep}.json"))
print(f"BIG_TILE={bt:3s} {d['value']:8.1f} frames/s  3-D {d['config']['ms_3d_step']:.2f} ms  2-D {d['config']['ms_2d_step']:.2f} ms  2-D frac {d['roofline_step_2d']['frac']:.4f}")
PY
done
done
