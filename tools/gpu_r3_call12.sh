#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
PMC_VARIANTS=3 bash tools/gpu_pmc.sh quad16 python $PWD/tools/lab_pmc_gemm.py 155648 4096 4096 > /dev/null
python tools/pmc_summary.py gpurun_out/pmc_quad16 "" > gpurun_out/pmc_quad16/summary.txt 2>&1
grep -A22 "quad\|MT256" gpurun_out/pmc_quad16/summary.txt | grep -v "^--" | head -50
rm -f gpurun_out/pmc_quad16/*.db
