python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gated_conv_train or wgrad or weight_prep" 2>&1 | tail -5
echo "--- stream (default)"; python scratch/c32_bench.py 2 2>&1 | grep -v amdgpu.ids
echo "--- tile kernel (ONIRIS_WGRAD=2)"; ONIRIS_WGRAD=2 python scratch/c32_bench.py 2 2>&1 | grep wgrad
echo "--- B=8 stream"; python scratch/c32_bench.py 8 2>&1 | grep wgrad
echo "--- B=8 tile"; ONIRIS_WGRAD=2 python scratch/c32_bench.py 8 2>&1 | grep wgrad
echo "--- 96->32 B=2 stream / tile"; python scratch/c32_bench.py 2 96 32 2>&1 | grep wgrad;  ONIRIS_WGRAD=2 python scratch/c32_bench.py 2 96 32 2>&1 | grep wgrad
