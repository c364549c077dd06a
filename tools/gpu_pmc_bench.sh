#!/bin/bash
# HBM traffic counters (separate passes, kernel-trace only) over one short bench run
# MODEL=clip-flant5-xl|clip-flant5-xxl (default xxl, the metric's model); output in gpurun_out/pmc_bench_<xl|xxl>/
MODEL=${MODEL:-clip-flant5-xxl}; TAG=${MODEL#clip-flant5-}; OUT=gpurun_out/pmc_bench_$TAG
REPO=$(pwd); mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS -d $REPO/$OUT -o pass$i -- python $REPO/bench.py --model $MODEL --steps 1 --warmup 1 --cpu-pairs 0 --also none > $REPO/$OUT/pass$i.log 2>&1
  echo "pass $i exit $?"
done
cd $REPO
python tools/pmc_summary.py $OUT vqs:: > $OUT/summary.txt 2>&1
rm -f $OUT/*.db
tail -5 $OUT/summary.txt
cp $OUT/gemm_traffic.json $OUT/gemm_traffic_${TAG}_b256.json
