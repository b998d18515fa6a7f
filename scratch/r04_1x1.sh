python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv_plain or conv1x1 or epilogue or attention" 2>&1 | tail -3
python scratch/conv1x1_bench.py 2>&1 | grep -v amdgpu
echo "--- previous build (full wait behind every epilogue)"
ONIRIS_LIB_NAME=liboniris_hip_old.so python scratch/conv1x1_bench.py 2>&1 | grep -v amdgpu
