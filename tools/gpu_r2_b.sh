#!/bin/bash
# Round-2 bench checks: default line (XXL + cpu_baseline), GenAI-Bench-1600 stand-in (bucketed) vs one ragged padded batch,
# and the self-launching N=2 path on one GPU (gloo, ranks share the device).
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
( time timeout 1200 python bench.py ) > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.log | cut -c1-300; grep real gpurun_out/bench_default.err
timeout 900 python bench.py --workload genai1600 --cpu-pairs 0 > gpurun_out/bench_genai1600.log 2> gpurun_out/bench_genai1600.err; tail -1 gpurun_out/bench_genai1600.log | cut -c1-400
timeout 600 python bench.py --ragged --cpu-pairs 0 --steps 3 --warmup 1 > gpurun_out/bench_ragged.log 2>&1; tail -1 gpurun_out/bench_ragged.log | cut -c1-200
VQS_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --model clip-flant5-xl --steps 3 --warmup 1 --cpu-pairs 0 > gpurun_out/bench_gpus2_gloo.log 2> gpurun_out/bench_gpus2_gloo.err; tail -1 gpurun_out/bench_gpus2_gloo.log | cut -c1-300; tail -3 gpurun_out/bench_gpus2_gloo.err
