#!/bin/bash
# First GPU call of the next round: the three measurements this round ended without.
#  1. kernel-trace summary of the Qwen2.5-VL cached decode step (where its 29.9 ms go: M = 64 GEMM launches vs one-row attention)
#  2. per-call-site GEMM table + per-step times of the XXL main leg on this box (the baseline any GEMM change is compared with)
#  3. wall time of the whole GPU suite in one process (pieces took 5.3 + 6.7 + 2.5 min at the end of round 3)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/next
REPO=$(pwd); export PYTHONUNBUFFERED=1
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/next -o qwen_decode -- \
    python $REPO/tools/bench_qwen.py --batch 64 --steps 1 --warmup 0 --decode-steps 16 > $REPO/gpurun_out/next/qwen_decode.log 2>&1 )
python tools/rocpd_summary.py gpurun_out/next/qwen_decode_results.db gpurun_out/next/qwen_decode_summary \
    "rocprofv3 --kernel-trace --stats -- python tools/bench_qwen.py --batch 64 --steps 1 --warmup 0 --decode-steps 16" > /dev/null 2>&1
rm -f gpurun_out/next/*.db
head -24 gpurun_out/next/qwen_decode_summary.md
VQS_BENCH_REPORT=1 timeout 120 python bench.py --gpus 1 --steps 8 --warmup 2 --cpu-pairs 0 --also none 2> gpurun_out/next/gemm_report.txt | tail -1 > gpurun_out/next/bench_main_leg.json
grep -v amdgpu gpurun_out/next/gemm_report.txt | head -24
S=$(date +%s); timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 2>&1 | tail -20 > gpurun_out/next/gpu_suite.txt; echo "gpu suite wall $(( $(date +%s) - S )) s"; tail -16 gpurun_out/next/gpu_suite.txt
