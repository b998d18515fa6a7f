echo "--- B=8 stream"; python scratch/c32_bench.py 8 2>&1 | grep conv_stream
echo "--- B=8 stream stagger"; ONIRIS_BIG_TILE=20 python scratch/c32_bench.py 8 2>&1 | grep conv_stream
echo "--- B=2 stream"; python scratch/c32_bench.py 2 2>&1 | grep conv_stream
echo "--- B=2 stream stagger"; ONIRIS_BIG_TILE=20 python scratch/c32_bench.py 2 2>&1 | grep conv_stream
