#!/bin/bash
# run the GEMM microbench (variant 0) against each lab library
mkdir -p gpurun_out; rm -f gpurun_out/lab.txt
for L in "" $(ls build/lab/*.so); do
  echo "== ${L:-product}" >> gpurun_out/lab.txt
  VQS_LIB_PATH=${L:-t2v_metrics_amd/libvqs_hip.so} VQS_BENCH_VARIANTS=${VQS_BENCH_VARIANTS:-0} python tools/microbench.py --gemm-only $LABARGS 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(f\"{r['tag']:14s} v{r['variant']} {r['ms']:8.3f} ms {r['tflops']:7.1f} TF\")
" >> gpurun_out/lab.txt
done
cat gpurun_out/lab.txt
