ONIRIS_BIG_TILE=3 ONIRIS_WGRAD=2 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gated_conv or conv_epilogue or conv_plain" 2>&1 | tail -2
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gated_conv" 2>&1 | tail -2
