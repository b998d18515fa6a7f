#!/bin/bash
# does measuring the rollout extras first (child processes, ~1 min of GPU work) change the training headline that follows?  same box, alternating
run() { echo -n "$1: "; python bench.py --cpu-frames 0 --no-profile $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'frames/s', d['config']['ms_3d_step'], d['config']['ms_2d_step'])"; }
run "--no-extra"; run ""; run "--no-extra"; run ""
