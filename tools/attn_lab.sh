#!/bin/bash
# A/B the attention microbench across lab libraries (build/lab/*.so) and the product library
mkdir -p gpurun_out; rm -f gpurun_out/attn_lab.txt
for L in "" $(ls build/lab/*.so); do
  echo "== ${L:-product}" >> gpurun_out/attn_lab.txt
  VQS_LIB_PATH=${L:-t2v_metrics_amd/libvqs_hip.so} python tools/microbench.py --no-gemm 2>&1 | grep attention | cut -c1-150 >> gpurun_out/attn_lab.txt
done
cat gpurun_out/attn_lab.txt
