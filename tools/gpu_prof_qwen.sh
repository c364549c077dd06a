#!/bin/bash
# Qwen2.5-VL-7B: bench line + rocprofv3 kernel stats
mkdir -p gpurun_out/prof_qwen
export PYTHONUNBUFFERED=1
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_qwen -o bench -- python $REPO/tools/bench_qwen.py --batch 32 --steps 2 --warmup 1 > $REPO/gpurun_out/prof_qwen/log.txt 2>&1
cd $REPO
python tools/rocpd_summary.py gpurun_out/prof_qwen/bench_results.db gpurun_out/prof_qwen/summary "rocprofv3 --kernel-trace --stats -- python tools/bench_qwen.py --batch 32 --steps 2 --warmup 1" > /dev/null 2>&1
rm -f gpurun_out/prof_qwen/*.db
tail -1 gpurun_out/prof_qwen/log.txt | cut -c1-300
head -30 gpurun_out/prof_qwen/summary.md
