python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gated_conv or conv_epilogue" 2>&1 | tail -3
echo "--- B=8 stream"; python scratch/c32_bench.py 8 2>&1 | grep conv_stream
echo "--- B=2 stream"; python scratch/c32_bench.py 2 2>&1 | grep conv_stream
ONIRIS_LIB_NAME=liboniris_hip_stamp.so python scratch/stream_stamp.py 8 2>&1 | grep -v amdgpu.ids | head -10
