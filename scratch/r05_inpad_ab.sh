#!/bin/bash
# packed UNet input 16 channels wide (rounds 1-4) against 32 (the stem conv and its weight gradient on the streaming kernels)
O=gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for rep in 1 2; do
for v in 16 32; do
  ONIRIS_IN_PAD=$v python bench.py --steps 12 --warmup 4 --cpu-frames 0 --no-extra > $O/ab6_${v}_$rep.json 2>/dev/null
  ONIRIS_IN_PAD=$v python bench.py --mode rollout --gen-frames 48 --batch 1 > $O/ab6_ro_${v}_$rep.json 2>/dev/null
  python - $v $rep <<'PY'
import json, sys
v, rep = sys.argv[1:]
d = json.load(open(f"gpurun_out/ab6_{v}_{rep}.json")); r = json.load(open(f"gpurun_out/ab6_ro_{v}_{rep}.json"))
print(f"IN_PAD={v:3s} {d['value']:8.1f} frames/s  3-D {d['config']['ms_3d_step']:.2f} ms  2-D {d['config']['ms_2d_step']:.2f} ms  step frac {d['roofline_step']['frac']:.4f} | rollout {r['value']:.2f} frames/s {r['ms_per_unet_eval']:.4f} ms/eval")
PY
done
done
