#!/bin/bash
# A/B the attention microbench: register-staged kernel (VQS_ATTN_VARIANT=0) vs LDS-DMA kernel (1), plus any lab
# libraries under build/lab/ (ablations built with -DVQS_ATTN_ABLATE=n)
mkdir -p gpurun_out; rm -f gpurun_out/attn_lab.txt
for V in 0 1; do
  echo "== product, VQS_ATTN_VARIANT=$V" >> gpurun_out/attn_lab.txt
  VQS_ATTN_VARIANT=$V python tools/microbench.py --no-gemm 2>&1 | grep attention | cut -c1-150 >> gpurun_out/attn_lab.txt
done
for L in $(ls build/lab/*.so 2>/dev/null); do
  echo "== $L" >> gpurun_out/attn_lab.txt
  VQS_LIB_PATH=$L python tools/microbench.py --no-gemm 2>&1 | grep attention | cut -c1-150 >> gpurun_out/attn_lab.txt
done
cat gpurun_out/attn_lab.txt
