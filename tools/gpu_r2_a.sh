#!/bin/bash
# Round-2 opening measurement of HEAD in one gpurun call: GPU tests, XXL bench + per-call-site GEMM report + rocprofv3
# kernel stats + PMC traffic passes (the metric's model), then the XL bench + kernel stats.  Output: gpurun_out/.
mkdir -p gpurun_out
bash tools/gpu_round.sh tests
bash tools/gpu_prof_xxl.sh 2>&1 | tail -40
MODEL=clip-flant5-xxl bash tools/gpu_pmc_bench.sh 2>&1 | tail -4
VQS_BENCH_REPORT=1 timeout 600 python bench.py --model clip-flant5-xl --steps 5 --warmup 2 --cpu-pairs 0 > gpurun_out/bench_xl.log 2> gpurun_out/gemm_report_xl.txt; tail -1 gpurun_out/bench_xl.log | cut -c1-200
