#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/gpu_pmc.sh gemm_forms python $PWD/tools/lab_pmc_gemm.py 155648 4096 4096
python tools/pmc_summary.py gpurun_out/pmc_gemm_forms "" > gpurun_out/pmc_gemm_forms/summary.txt 2>&1
grep -v "^    .*n=1," gpurun_out/pmc_gemm_forms/summary.txt | head -150
rm -f gpurun_out/pmc_gemm_forms/*.db
