#!/bin/bash
# usage: flaky.sh <n> <extra bench args...>
n=$1; shift
for k in $(seq 1 $n); do
  python bench.py --steps 8 --warmup 4 --cpu-frames 0 --no-profile "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['loss'],3))"
done
