#!/bin/bash
# One GPU call: (1) the new tile-order GPU test on the product library, (2) tools/lab_call.py (tile-order sweep, in-situ A/B,
# attention bias-in-accumulator A/B), (3) the attention / stage-locked / noise-floor parity tests on the VARIANT library.
# Logs under gpurun_out/.
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl gpurun_out/lab_*.log
timeout 240 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "tile_order" -x > gpurun_out/lab_pytest_tile_order.log 2>&1
echo "pytest tile_order exit $?" >> gpurun_out/lab_pytest_tile_order.log
timeout 330 python tools/lab_call.py --parts A,C,B,D,X > gpurun_out/lab_call.log 2>&1
echo "lab_call exit $?" >> gpurun_out/lab_call.log
VQS_LIB_PATH=$PWD/build/lab/libvqs_attn_bias_acc.so timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stage_locked.py tests/test_gpu_parity_noise_floor.py \
  -q -m gpu -k "attention or tiny or small or fixtures or ragged_batches" > gpurun_out/lab_pytest_variant.log 2>&1
echo "pytest variant exit $?" >> gpurun_out/lab_pytest_variant.log
tail -3 gpurun_out/lab_pytest_tile_order.log; tail -25 gpurun_out/lab_call.log; tail -8 gpurun_out/lab_pytest_variant.log
