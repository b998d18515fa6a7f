#!/bin/bash
# two K-groups per workgroup (default) against two independent 4-wave workgroups per CU (make variant VSRC=conv_wgrad VNAME=wng1 VDEF=-DWGRAD_NG=1)
O=gpurun_out
ONIRIS_LIB_NAME=liboniris_hip_wng1c.so ONIRIS_WGRAD_CAP_MUL=2 python -m pytest tests -m gpu -q -x -k "wgrad or gated_conv_train or g8_" 2>&1 | tail -3
for rep in 1 2; do
for cfg in "liboniris_hip_wng1b.so 2" "liboniris_hip_wng1c.so 2"; do
  set -- $cfg
  ONIRIS_LIB_NAME=$1 ONIRIS_WGRAD_CAP_MUL=$2 python bench.py --steps 12 --warmup 4 --cpu-frames 0 --no-extra > $O/ab_wng_$1_$rep.json 2> $O/ab_wng_$1_$rep.err
  python - $1 $rep <<'PY'
import json, sys
lib, rep = sys.argv[1:]
d = json.load(open(f"gpurun_out/ab_wng_{lib}_{rep}.json"))
k = d["kernels"]
w = {n[:30]: (v["ms_total"], v["roof"]) for n, v in k.items() if n.startswith("conv_wgrad")}
print(f"{lib:26s} {d['value']:8.1f} frames/s  3-D {d['config']['ms_3d_step']:.2f} ms  2-D {d['config']['ms_2d_step']:.2f} ms  step frac {d['roofline_step']['frac']:.4f}  wgrad_glds {w}")
PY
done
done
