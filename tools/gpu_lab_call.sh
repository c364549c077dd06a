#!/bin/bash
# One GPU call of tools/lab_call.py (parts in $1, default H,N,F,X) + the GEMM GPU tests on the product library.  Logs: gpurun_out/.
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl gpurun_out/lab_*.log
timeout 400 python tools/lab_call.py --parts "${1:-H,N,F,X}" > gpurun_out/lab_call.log 2>&1
echo "lab_call exit $?" >> gpurun_out/lab_call.log
timeout 240 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" -x > gpurun_out/lab_pytest_gemm.log 2>&1
echo "pytest gemm exit $?" >> gpurun_out/lab_pytest_gemm.log
tail -40 gpurun_out/lab_call.log | cut -c1-700; tail -4 gpurun_out/lab_pytest_gemm.log
