#!/bin/bash
# (needs the lab build: make -C t2v_metrics_amd/csrc lab -- the shipped library reads no environment variables)
# A/B of the deferred stream store (VQS_NORM_DEFER) on one box: GPU tests, then the bench with the store in every norm
# and with the default, twice each and interleaved.  Output: gpurun_out/ab_norm.log
bash tools/gpu_round.sh tests
grep -E "FAILED|ERROR" gpurun_out/pytest_gpu.log | head -10
: > gpurun_out/ab_norm.log
for rep in 1 2; do
  for mode in 0 1; do
    echo "VQS_NORM_DEFER=$mode rep $rep" >> gpurun_out/ab_norm.log
    VQS_LIB_PATH=build/lab/libvqs_hip_lab.so VQS_NORM_DEFER=$mode VQS_BENCH_REPORT=1 timeout 600 python bench.py --steps 4 --warmup 1 --cpu-pairs 0 >> gpurun_out/ab_norm.log 2> gpurun_out/ab_norm_report_${mode}.txt
  done
done
python - <<'PY'
import json
for line in open('gpurun_out/ab_norm.log'):
    if line.startswith('{'):
        j = json.loads(line); print(round(j['value'], 1), 'pairs/s', round(j['ms_per_step'], 1), 'ms', round(j['roofline']['achieved'], 1), 'TF')
    else:
        print(line.strip())
PY
grep -i "norm" gpurun_out/ab_norm_report_0.txt | head -8
grep -i "norm" gpurun_out/ab_norm_report_1.txt | head -8
