#!/bin/bash
# HBM traffic counters (separate passes, kernel-trace only) over one short bench run
REPO=$(pwd); mkdir -p gpurun_out/pmc_bench; cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS -d $REPO/gpurun_out/pmc_bench -o pass$i -- python $REPO/bench.py --steps 1 --warmup 1 --cpu-pairs 0 > $REPO/gpurun_out/pmc_bench/pass$i.log 2>&1
  echo "pass $i exit $?"
done
cd $REPO
python tools/pmc_summary.py gpurun_out/pmc_bench vqs:: > gpurun_out/pmc_bench/summary.txt 2>&1
rm -f gpurun_out/pmc_bench/*.db
tail -5 gpurun_out/pmc_bench/summary.txt
