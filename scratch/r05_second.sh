#!/bin/bash
# round 5, second GPU call: scalar-gradient measurement (incl. the slow full-size oracle runs), the changed tests, the new bench
# records (roofline_step_2d, extra.cs_t64, extra.rollout_256) and what reserving CUs for RCCL costs on ONE GPU
O=gpurun_out
python -m pytest tests/test_model_gpu.py -m gpu -q -x -s -k "cs_shaped_unet_vs_oracle or g6 or ddp" > $O/r05_scalar_grads.log 2>&1; echo rc=$? >> $O/r05_scalar_grads.log
grep -E "scalar gradients|passed|failed|rc=" $O/r05_scalar_grads.log
python -m pytest tests/test_ops_gpu.py tests/test_abi.py -m gpu -q -x > $O/r05_ops.log 2>&1; echo rc=$? >> $O/r05_ops.log; tail -3 $O/r05_ops.log
python bench.py --steps 12 --warmup 4 --cpu-frames 0 > $O/r05_bench_2.json 2> $O/r05_bench_2.err; tail -c 600 $O/r05_bench_2.err
for k in 8 16 32; do
ONIRIS_COMM_CUS_ALWAYS=$k python bench.py --steps 12 --warmup 4 --cpu-frames 0 --no-extra --no-profile > $O/r05_bench_cus$k.json 2> $O/r05_bench_cus$k.err
done
python bench.py --steps 12 --warmup 4 --cpu-frames 0 --no-extra --no-profile > $O/r05_bench_cus0.json 2> $O/r05_bench_cus0.err
python - <<'PY'
import json
for k in (0, 8, 16, 32):
    d = json.load(open(f"gpurun_out/r05_bench_cus{k}.json"))
    print("reserve", k, "frames/s", round(d["value"], 1), "ms 3d/2d", d["config"]["ms_3d_step"], d["config"]["ms_2d_step"])
d = json.load(open("gpurun_out/r05_bench_2.json"))
print(d["value"], {k: (v.get("frames_s") or v.get("value")) for k, v in d["extra"].items() if isinstance(v, dict)}, d["extra"].get("error"))
print(d["roofline_step_2d"])
PY
