#!/bin/bash
# Profiles of HEAD for the metric's configuration (clip-flant5-xxl, B = 256) in one call: rocprofv3 kernel-trace summary +
# by-grid split, the per-call-site GEMM table, the FETCH_SIZE / WRITE_SIZE PMC passes (HBM-side bytes per GEMM launch).
export PYTHONUNBUFFERED=1
MODEL=clip-flant5-xxl bash tools/gpu_prof.sh 2>&1 | tail -2
VQS_BENCH_REPORT=1 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-pairs 0 > gpurun_out/bench_xxl.log 2> gpurun_out/gemm_report_xxl.txt
tail -1 gpurun_out/bench_xxl.log | cut -c1-300
OUT=gpurun_out/pmc_bench_xxl; REPO=$(pwd); mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $CTRS -d $REPO/$OUT -o pass$i -- python $REPO/bench.py --steps 1 --warmup 1 --cpu-pairs 0 > $REPO/$OUT/pass$i.log 2>&1
  echo "pmc pass $i exit $?"
done
cd $REPO
python tools/pmc_summary.py $OUT vqs:: > $OUT/summary.txt 2>&1
rm -f $OUT/*.db
grep -A3 "gemm_bf16_persistent<5" $OUT/summary.txt | head -8
head -14 gpurun_out/prof_xxl/summary.md
