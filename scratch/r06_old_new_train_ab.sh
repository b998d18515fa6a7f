#!/bin/bash
# same-box A/B of the training bench: library of commit a6bc474 (before the round's evaluation-kernel work; built as liboniris_hip_old.so) vs HEAD
run() { echo -n "$1: "; env $1 python bench.py --steps 10 --warmup 4 --cpu-frames 0 --no-extra --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'frames/s', d['config']['ms_3d_step'], d['config']['ms_2d_step'])"; }
for i in 1 2 3; do run "ONIRIS_LIB_NAME=liboniris_hip_old.so"; run "X=1"; done
