#!/bin/bash
# round 5, first GPU call: the suite on the new tree + the default bench (this round's starting point)
O=gpurun_out
python -m pytest tests -m gpu -q -x > $O/r05_gputests_1.log 2>&1; echo rc=$? >> $O/r05_gputests_1.log
tail -15 $O/r05_gputests_1.log
python bench.py --steps 12 --warmup 4 > $O/r05_bench_start.json 2> $O/r05_bench_start.err
tail -c 1500 $O/r05_bench_start.json
