#!/bin/bash
# XXL B=256: bench line + per-call-site GEMM report + rocprofv3 kernel stats
mkdir -p gpurun_out/prof_xxl
export PYTHONUNBUFFERED=1
VQS_BENCH_REPORT=1 timeout 900 python bench.py --steps 3 --warmup 1 --model clip-flant5-xxl --cpu-pairs 0 > gpurun_out/bench_xxl.log 2> gpurun_out/gemm_report_xxl.txt
tail -1 gpurun_out/bench_xxl.log | cut -c1-250
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_xxl -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --model clip-flant5-xxl --cpu-pairs 0 > $REPO/gpurun_out/prof_xxl/log.txt 2>&1
cd $REPO
python tools/rocpd_summary.py gpurun_out/prof_xxl/bench_results.db gpurun_out/prof_xxl/summary "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --model clip-flant5-xxl --cpu-pairs 0" > /dev/null 2>&1
rm -f gpurun_out/prof_xxl/*.db
head -24 gpurun_out/prof_xxl/summary.md
grep -v amdgpu gpurun_out/gemm_report_xxl.txt
