python -m pytest tests/test_ops_gpu.py -m gpu -q -x -s -k "conv_epilogues" 2>&1 | grep -v "^$" | tail -16
python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "g7 or g8 or cs_shaped" 2>&1 | tail -3
for m in 1 0; do ONIRIS_CLIP_FLAG=$m python bench.py --steps 12 --warmup 4 --cpu-frames 0 --no-extra --no-profile > gpurun_out/r04_clip_$m.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r04_clip_$m.json'));print('clip_flag=$m', round(d['value']), round(d['ms_per_step'],2), d['config']['ms_3d_step'], d['config']['ms_2d_step'])"; done
for m in 1 0; do ONIRIS_CLIP_FLAG=$m python bench.py --steps 12 --warmup 4 --cpu-frames 0 --no-extra --no-profile > gpurun_out/r04_clip_$m.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r04_clip_$m.json'));print('clip_flag=$m', round(d['value']), round(d['ms_per_step'],2), d['config']['ms_3d_step'], d['config']['ms_2d_step'])"; done
