#!/bin/bash
# In-situ A/B of two library builds on one box: build/lab/libvqs_ab_old.so vs build/lab/libvqs_ab_new.so, interleaved,
# bench.py (XXL default, or MODEL=...) with the per-call-site GEMM report.  Output: gpurun_out/ab_lib.log
mkdir -p gpurun_out; rm -f gpurun_out/ab_lib.log
MODEL=${MODEL:-clip-flant5-xxl}
for rep in 1 2; do
  for which in old new; do
    echo "== $which rep $rep" >> gpurun_out/ab_lib.log
    VQS_LIB_PATH=build/lab/libvqs_ab_$which.so VQS_BENCH_REPORT=1 timeout 600 python bench.py --model $MODEL --steps 4 --warmup 1 --cpu-pairs 0 2> gpurun_out/ab_lib_report_${which}_$rep.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/ab_lib.log
    grep -E "enc o |enc wo|enc qkv|enc wi|vit" gpurun_out/ab_lib_report_${which}_$rep.txt >> gpurun_out/ab_lib.log
  done
done
cat gpurun_out/ab_lib.log
