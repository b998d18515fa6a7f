#!/bin/bash
# sample power / clocks while the bench runs (is the step power-bound?)
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|Max" | head -8
python bench.py --steps 400 --warmup 5 --cpu-frames 0 --no-profile "$@" > /tmp/b.json 2>/dev/null &
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|junction|Power \(W\)" | tr '\n' ' '; echo
  sleep 1
done
wait $PID
python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"
