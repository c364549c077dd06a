#!/bin/bash
# L2 / fabric counters of the XXL encoder's GEMM shapes, quad form and hipBLASLt side by side (tools/lab_gemm_wo.py); output gpurun_out/<tag>/pmc_wo/
TAG=${1:-wo}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG/pmc_wo; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "GRBM_GUI_ACTIVE FETCH_SIZE" \
            "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT -o pass$i -- python $REPO/tools/lab_gemm_wo.py P > $OUT/pass$i.log 2>&1
  echo "pass $i ($CTRS) exit $?"
done
cd $REPO
python tools/pmc_summary.py gpurun_out/$TAG/pmc_wo "gemm_bf16_quad|Cijk" --each > $OUT/summary.txt 2>&1; grep -c . $OUT/summary.txt
python tools/lab_gemm_wo.py T > $OUT/tile_order_sweep.jsonl 2>&1; tail -3 $OUT/tile_order_sweep.jsonl | cut -c1-1500
rm -f $OUT/*.db
