#!/bin/bash
# the reference loop as written, without and with torch DistributedDataParallel (one RCCL rank), and the GPU tests of that path
O=gpurun_out
python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "ddp or reference_training_loop or fused_dart" 2>&1 | tail -2
python bench.py --wrapper torch --steps 8 --warmup 4 --cpu-frames 0 --no-extra --no-profile > $O/r05_torchloop_1gpu.json 2> $O/r05_torchloop_1gpu.err
ONIRIS_FORCE_DIST=1 python bench.py --wrapper torch --steps 8 --warmup 4 --cpu-frames 0 --no-extra --no-profile > $O/r05_torchddp_1rank.json 2> $O/r05_torchddp_1rank.err
ONIRIS_FORCE_DIST=1 python bench.py --steps 8 --warmup 4 --cpu-frames 0 --no-extra --no-profile > $O/r05_onirisddp_1rank.json 2> $O/r05_onirisddp_1rank.err
python3 - <<'PY'
import json
for f in ("r05_torchloop_1gpu", "r05_torchddp_1rank", "r05_onirisddp_1rank"):
    try:
        d = json.load(open("gpurun_out/" + f + ".json"))
        print(f, round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step  3-D", d["config"]["ms_3d_step"], "2-D", d["config"]["ms_2d_step"], (d.get("ddp") or {}).get("kernel_owned_parameters"))
    except Exception as e:
        print(f, "ERR", e, open("gpurun_out/" + f + ".err").read()[-1500:])
PY
