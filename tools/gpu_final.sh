#!/bin/bash
# Round-end measurement in one gpurun call: GPU tests, smoke, bench (with CPU baseline), rocprofv3 kernel stats,
# PMC traffic passes.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
bash tools/gpu_round.sh tests
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.log | cut -c1-260
bash tools/gpu_prof.sh 2>&1 | tail -2
if [ "$1" != "nopmc" ]; then          # the three --pmc passes take ~2.5 min; skip when the GEMM kernels did not change
  bash tools/gpu_pmc_bench.sh 2>&1 | tail -3
  cp gpurun_out/pmc_bench/gemm_traffic.json gpurun_out/ 2>/dev/null
fi
head -14 gpurun_out/prof/summary.md
