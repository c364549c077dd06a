#!/bin/bash
# Round measurement of HEAD in one gpurun call: GPU tests, smoke, the default bench line (XXL + cpu_baseline), rocprofv3
# kernel stats + by-grid split + PMC passes for XXL (the metric's model) and XL.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash tools/gpu_round.sh tests
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-260
for MODEL in clip-flant5-xxl clip-flant5-xl; do
  TAG=${MODEL#clip-flant5-}
  MODEL=$MODEL bash tools/gpu_prof.sh 2>&1 | tail -2
  VQS_BENCH_REPORT=1 timeout 900 python bench.py --model $MODEL --steps 3 --warmup 1 --cpu-pairs 0 > gpurun_out/bench_$TAG.log 2> gpurun_out/gemm_report_$TAG.txt
  if [ "$1" != "nopmc" ]; then MODEL=$MODEL bash tools/gpu_pmc_bench.sh 2>&1 | tail -3; fi
  head -16 gpurun_out/prof_$TAG/summary.md
done
