python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "g7_blocks or g8_unet_loss or cs_shaped or g9b" 2>&1 | grep -v "^$" | cut -c1-600 > gpurun_out/r04_tol.log
python -m pytest tests/test_ops_gpu.py -m gpu -q -s -k "gated_conv_train" 2>&1 | grep -v "^$" | cut -c1-400 >> gpurun_out/r04_tol.log
tail -60 gpurun_out/r04_tol.log
