O=gpurun_out
ONIRIS_FORCE_DIST=1 python bench.py --steps 4 --warmup 2 --cpu-frames 0 --no-extra --batch 2 > $O/r04_forcedist.json 2> $O/r04_forcedist.err; echo rc=$?; tail -2 $O/r04_forcedist.err
python -c "
import json;d=json.load(open('gpurun_out/r04_forcedist.json'));print(round(d['value']), d['backend'], d['rccl_world'], d['ddp'], d['device_of_rank'])"
for ex in allreduce mesh; do
ONIRIS_DDP_EXCHANGE=$ex ONIRIS_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 4 --warmup 2 --cpu-frames 0 --no-profile --batch 2 > $O/r04_share2_$ex.json 2> $O/r04_share2_$ex.err; echo rc=$?; tail -2 $O/r04_share2_$ex.err | cut -c1-300
python -c "
import json;d=json.load(open('gpurun_out/r04_share2_$ex.json'));print(round(d['value']), d['backend'], d['rccl_world'], d['ddp'], d['device_of_rank'], d['config'].get('shared_gpu_gloo_NOT_A_MEASUREMENT'))"
done
ONIRIS_DDP_EXCHANGE=mesh ONIRIS_DDP_BF16=1 ONIRIS_FORCE_DIST=1 python bench.py --steps 4 --warmup 2 --cpu-frames 0 --no-extra --batch 2 --no-profile > $O/r04_forcedist_mesh.json 2> $O/r04_forcedist_mesh.err; echo rc=$?
python -c "
import json;d=json.load(open('gpurun_out/r04_forcedist_mesh.json'));print(round(d['value']), d['backend'], d['rccl_world'], d['ddp'])"
