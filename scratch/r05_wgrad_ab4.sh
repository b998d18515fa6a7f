#!/bin/bash
# conv_wgrad_stream_kernel: round-4 form (wsold) | + halo-row order (wsph8) | + 4x16-pixel tiles, two workgroups per CU (default build)
O=gpurun_out
python -m pytest tests -m gpu -q -x -k "wgrad or gated_conv_train or g8_ or g3_" 2>&1 | tail -3
for rep in 1 2; do
for lib in liboniris_hip_wsold.so liboniris_hip_wsph8.so liboniris_hip.so; do
  ONIRIS_LIB_NAME=$lib python bench.py --steps 12 --warmup 4 --cpu-frames 0 --no-extra > $O/ab4_${lib}_$rep.json 2>/dev/null
  python - $lib $rep <<'PY'
import json, sys
lib, rep = sys.argv[1:]
d = json.load(open(f"gpurun_out/ab4_{lib}_{rep}.json"))
k = d["kernels"]
w = {n[:28]: (v["ms_total"], v["roof"]) for n, v in k.items() if "wgrad_stream" in n}
print(f"{lib:26s} {d['value']:8.1f} frames/s  3-D {d['config']['ms_3d_step']:.2f} ms  2-D {d['config']['ms_2d_step']:.2f} ms  step frac {d['roofline_step']['frac']:.4f}  {w}")
PY
done
done
