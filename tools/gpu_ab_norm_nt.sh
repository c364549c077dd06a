#!/bin/bash
# In-situ A/B of non-temporal stream accesses in the norm kernels (lab libraries from tools/norm_nt_lab.sh <bits>; 0 here = the in-tree default library)
mkdir -p gpurun_out; : > gpurun_out/ab_norm_nt.log
for rep in 1 2; do
  for n in 0 7; do
    lib=t2v_metrics_amd/libvqs_hip.so; [ $n != 0 ] && lib=build/lab/libvqs_nnt$n.so
    echo "bench VQS_NORM_NT=$n" >> gpurun_out/ab_norm_nt.log
    VQS_LIB_PATH=$PWD/$lib timeout 600 python bench.py --steps 4 --warmup 1 --cpu-pairs 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1),'pairs/s', round(j['ms_per_step'],2),'ms', round(j['roofline']['achieved'],1), 'TF')" >> gpurun_out/ab_norm_nt.log
  done
done
cat gpurun_out/ab_norm_nt.log
