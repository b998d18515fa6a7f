#!/bin/bash
# same-box A/B of conv_glds role / priority variants (make variant ...): scratch/mid_bench.py under each build, twice
for rep in 1 2; do
for v in "" _ey _pl6 _pl12 _pl18 _alt _eyalt; do
  ONIRIS_LIB_NAME=liboniris_hip$v.so python scratch/mid_bench.py 8 2>&1 | tail -1
done; done
