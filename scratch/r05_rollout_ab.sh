#!/bin/bash
O=gpurun_out
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "cat_act or resample" 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py tests/test_consistency_gpu.py -m gpu -q -x -k "sampler or g9 or rollout or cached or eval or consistency" 2>&1 | tail -2
for v in 0 1 0 1; do
  ONIRIS_CAT_ACT_FUSED=$v python bench.py --mode rollout --gen-frames 48 --batch 1 > $O/ro_cat$v.json 2> $O/ro_cat$v.err
  python3 -c "import json; d=json.load(open('$O/ro_cat$v.json')); print('cat_act_fused=$v', round(d['value'],2), 'frames/s', round(d['ms_per_unet_eval'],4), 'ms/eval', d['finite'])"
done
