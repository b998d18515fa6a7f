#!/bin/bash
O=gpurun_out
python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_consistency_gpu.py -m gpu -q -x > $O/r05_gputests_alias.log 2>&1; echo rc=$? >> $O/r05_gputests_alias.log; tail -3 $O/r05_gputests_alias.log
bash scratch/r05_ab.sh ONIRIS_ALIAS2 0 1
bash scratch/r05_ab.sh ONIRIS_ALIAS2 0 1
