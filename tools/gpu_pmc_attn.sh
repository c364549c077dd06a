#!/bin/bash
# PMC passes over the attention microbench (LDS conflicts, wait breakdown, fabric fetch, L2 hit rate)
REPO=$(pwd); mkdir -p gpurun_out/pmc_attn; cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VMEM" \
            "GRBM_GUI_ACTIVE FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS -d $REPO/gpurun_out/pmc_attn -o pass$i -- python $REPO/tools/microbench.py --no-gemm > $REPO/gpurun_out/pmc_attn/pass$i.log 2>&1
  echo "pass $i exit $?"
done
cd $REPO
python tools/pmc_summary.py gpurun_out/pmc_attn attn_fwd > gpurun_out/pmc_attn/summary.txt 2>&1
rm -f gpurun_out/pmc_attn/*.db
cat gpurun_out/pmc_attn/summary.txt
