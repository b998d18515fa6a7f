python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gated_conv or conv_epilogue or conv_plain" 2>&1 | tail -8
echo "--- stream (default)"; python scratch/c32_bench.py 2 2>&1 | grep -v amdgpu.ids
echo "--- glds (ONIRIS_BIG_TILE=3)"; ONIRIS_BIG_TILE=3 python scratch/c32_bench.py 2 2>&1 | grep conv_glds
echo "--- B=8 stream"; python scratch/c32_bench.py 8 2>&1 | grep -v amdgpu.ids
echo "--- B=8 glds"; ONIRIS_BIG_TILE=3 python scratch/c32_bench.py 8 2>&1 | grep conv_glds
