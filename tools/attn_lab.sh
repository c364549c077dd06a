#!/bin/bash
# A/B the attention microbench: the shipped library, plus any lab
# libraries under build/lab/ (ablations built with -DVQS_ATTN_ABLATE=n)
mkdir -p gpurun_out; rm -f gpurun_out/attn_lab.txt
echo "== product" >> gpurun_out/attn_lab.txt
python tools/microbench.py --no-gemm 2>&1 | grep attention | cut -c1-150 >> gpurun_out/attn_lab.txt
for L in $(ls build/lab/libvqs_attn_*.so 2>/dev/null); do
  echo "== $L" >> gpurun_out/attn_lab.txt
  VQS_LIB_PATH=$L python tools/microbench.py --no-gemm 2>&1 | grep attention | cut -c1-150 >> gpurun_out/attn_lab.txt
done
cat gpurun_out/attn_lab.txt
