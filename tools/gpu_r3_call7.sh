#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/gpu_pmc.sh quad python $PWD/tools/lab_pmc_gemm.py 155648 4096 4096 > /dev/null
python tools/pmc_summary.py gpurun_out/pmc_quad "" > gpurun_out/pmc_quad/summary.txt 2>&1
grep -A22 "quad\|MT256\|persistent" gpurun_out/pmc_quad/summary.txt | grep -v "^--" | head -80
rm -f gpurun_out/pmc_quad/*.db
