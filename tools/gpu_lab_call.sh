#!/bin/bash
# One GPU call: the GEMM GPU tests on the product library, then tools/lab_call.py (parts in $1).  Logs: gpurun_out/.
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl gpurun_out/lab_*.log
timeout 240 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" -x > gpurun_out/lab_pytest_gemm.log 2>&1
echo "pytest gemm exit $?" >> gpurun_out/lab_pytest_gemm.log
tail -15 gpurun_out/lab_pytest_gemm.log
timeout 400 python tools/lab_call.py --parts "${1:-W,WI}" > gpurun_out/lab_call.log 2>&1
echo "lab_call exit $?" >> gpurun_out/lab_call.log
tail -40 gpurun_out/lab_call.log | cut -c1-600
