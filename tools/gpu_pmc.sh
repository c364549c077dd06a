#!/bin/bash
# PMC passes (separate runs, kernel-trace only) over a short GEMM microbench.  usage: gpu_pmc.sh <tag> "<python cmd>"
TAG=$1; shift
REPO=$(pwd)
mkdir -p gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
            "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS -d $REPO/gpurun_out/pmc_$TAG -o pass$i -- "$@" > $REPO/gpurun_out/pmc_$TAG/pass$i.log 2>&1
  echo "pass $i ($CTRS) exit $?"
done
cd $REPO; ls -la gpurun_out/pmc_$TAG | head
