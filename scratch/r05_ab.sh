#!/bin/bash
# same-box A/B of an environment knob over the default bench:  bash scratch/r05_ab.sh NAME VAL1 VAL2 ...
O=gpurun_out
name=$1; shift
for v in "$@"; do
  env $name=$v python bench.py --steps 12 --warmup 4 --cpu-frames 0 --no-extra > $O/ab_${name}_$v.json 2> $O/ab_${name}_$v.err
done
python - "$name" "$@" <<'PY'
import json, sys
name, vals = sys.argv[1], sys.argv[2:]
for v in vals:
    d = json.load(open(f"gpurun_out/ab_{name}_{v}.json"))
    ew = {k: x for k, x in d["kernels"].items()}
    print(f"{name}={v:>4}: {d['value']:8.1f} frames/s   3-D {d['config']['ms_3d_step']:.2f} ms  2-D {d['config']['ms_2d_step']:.2f} ms  step frac {d['roofline_step']['frac']:.4f}")
PY
