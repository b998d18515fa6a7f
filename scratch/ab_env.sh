#!/bin/bash
# same-box A/B of one environment knob: ab_env.sh VAR v1 v2 [v3 ...]   (two alternating rounds)
VAR=$1; shift
for r in 1 2; do for v in "$@"; do env $VAR=$v python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', round(d['value'],1), round(d['ms_per_step'],3))"; done; done
