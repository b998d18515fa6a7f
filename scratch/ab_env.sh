#!/bin/bash
# usage: ab_env.sh VAR v1 v2 ... : bench.py (no profile) once per value, twice round-robin
VAR=$1; shift
for rep in 1 2; do for v in "$@"; do
  export $VAR=$v
  python bench.py --cpu-frames 0 --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value'],1), round(d['ms_per_step'],3), d['loss'])"
done; done
