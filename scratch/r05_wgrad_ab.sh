#!/bin/bash
# halo-row order of the wgrad loop (default build) against the (k-step, tap) order (make variant VSRC=conv_wgrad VNAME=wro0 VDEF=-DWGRAD_ROWORDER=0)
O=gpurun_out
python -m pytest tests -m gpu -q -x -k "wgrad or gated_conv_train or conv_epilogues or g3_ or g7_ or g8_" 2>&1 | tail -3
for rep in 1 2; do
for lib in liboniris_hip_wro0.so liboniris_hip.so; do
  ONIRIS_LIB_NAME=$lib python bench.py --steps 12 --warmup 4 --cpu-frames 0 --no-extra > $O/ab_wro_${lib}_$rep.json 2> $O/ab_wro_${lib}_$rep.err
  python - $lib $rep <<'PY'
import json, sys
lib, rep = sys.argv[1:]
d = json.load(open(f"gpurun_out/ab_wro_{lib}_{rep}.json"))
k = d["kernels"]
w = {n.split("<")[1][:22]: (v["ms_total"], v["roof"]) for n, v in k.items() if n.startswith("conv_wgrad_glds")}
print(f"{lib:26s} {d['value']:8.1f} frames/s  3-D {d['config']['ms_3d_step']:.2f} ms  2-D {d['config']['ms_2d_step']:.2f} ms  step frac {d['roofline_step']['frac']:.4f}  wgrad_glds {w}")
PY
done
done
