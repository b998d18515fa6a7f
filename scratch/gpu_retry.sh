#!/bin/bash
# usage: scratch/gpu_retry.sh <timeout> '<command>'   -- retries while gpurun reports "no slot" (exit 3)
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
