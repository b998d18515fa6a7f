#!/bin/bash
# NG = 1 (default build) against NG = 2 (make variant VSRC=conv_wgrad VNAME=wng2 VDEF="-DWGRAD_NG=2 -DWGRAD_NG1=2") at the other configurations
O=gpurun_out
for rep in 1 2; do
for lib in liboniris_hip_wng2.so liboniris_hip.so; do
  ONIRIS_LIB_NAME=$lib python bench.py --batch 2 --steps 20 --warmup 5 --cpu-frames 0 --no-extra --no-profile > $O/ab3_b2_${lib}_$rep.json 2>/dev/null
  ONIRIS_LIB_NAME=$lib python bench.py --net cs --steps 8 --warmup 4 --no-profile --cpu-frames 0 > $O/ab3_cs_${lib}_$rep.json 2>/dev/null
  ONIRIS_LIB_NAME=$lib python bench.py --net cs --frames 64 --steps 8 --warmup 4 --no-profile --cpu-frames 0 > $O/ab3_cs64_${lib}_$rep.json 2>/dev/null
  python - $lib $rep <<'PY'
import json, sys
lib, rep = sys.argv[1:]
out = []
for c in ("b2", "cs", "cs64"):
    d = json.load(open(f"gpurun_out/ab3_{c}_{lib}_{rep}.json"))
    out.append(f"{c} {d['value']:8.1f} f/s (3-D {d['config']['ms_3d_step']:.2f} ms, 2-D {d['config']['ms_2d_step']:.2f} ms)")
print(f"{lib:26s}", " | ".join(out))
PY
done
done
