#!/bin/bash
# (needs the lab build: make -C t2v_metrics_amd/csrc lab -- the shipped library reads no environment variables)
# In-situ A/B on one box: L2-touch variant of the lock-step GEMM (VQS_L2_TOUCH) and lock-step forced everywhere
# (VQS_GEMM_VARIANT=7) against the defaults.  Output: gpurun_out/ab_touch.log
mkdir -p gpurun_out; : > gpurun_out/ab_touch.log
run() {
  echo "bench touch=$1 variant=$2" >> gpurun_out/ab_touch.log
  VQS_LIB_PATH=build/lab/libvqs_hip_lab.so VQS_L2_TOUCH=$1 VQS_GEMM_VARIANT=$2 VQS_BENCH_REPORT=1 timeout 600 python bench.py --steps 4 --warmup 1 --cpu-pairs 0 2> gpurun_out/ab_touch_report_$1_$2.txt | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1),'pairs/s', round(j['roofline']['achieved'],1),'TF')" >> gpurun_out/ab_touch.log
}
for rep in 1 2; do
  run 4 3; run 6 3; run 7 3; run 8 3
done
cat gpurun_out/ab_touch.log
for f in 4_3 6_3 7_3 8_3; do echo "== $f"; grep -E "vit|enc " gpurun_out/ab_touch_report_$f.txt; done
