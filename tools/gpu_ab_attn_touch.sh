#!/bin/bash
# In-situ A/B of the attention kernel's L2 touch (VQS_ATTN_TOUCH) on one box + the attention parity tests with it on.
mkdir -p gpurun_out; : > gpurun_out/ab_attn_touch.log
VQS_ATTN_TOUCH=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention" -p no:cacheprovider 2>&1 | tail -2 >> gpurun_out/ab_attn_touch.log
for rep in 1 2; do
  for mode in 0 1; do
    echo "bench VQS_ATTN_TOUCH=$mode" >> gpurun_out/ab_attn_touch.log
    VQS_ATTN_TOUCH=$mode timeout 600 python bench.py --steps 4 --warmup 1 --cpu-pairs 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1),'pairs/s', round(j['ms_per_step'],2),'ms', round(j['roofline']['gemm_share_of_step_time'],4))" >> gpurun_out/ab_attn_touch.log
  done
done
cat gpurun_out/ab_attn_touch.log
