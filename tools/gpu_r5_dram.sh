#!/bin/bash
# Round 5, VERDICT r4 item 3 (last sentence): can this rocprofv3 separate HBM bytes from Infinity-Cache bytes behind the fabric-side FETCH_SIZE?
# 1. lists the counters the tool offers and keeps the DRAM / MALL / EA ones; 2. collects those that exist over one bench step, one small pass
# each (kernel-trace + pmc only); 3. re-runs the attribution row of what ships (the dec_fp16 token was dropped by a bug in the tool's parser).
OUT=gpurun_out/r5dram; REPO=$(pwd); mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
(rocprofv3 --list-avail 2>&1 || rocprofv3-avail list 2>&1) > $REPO/$OUT/avail.txt
grep -o -E "\b(TCC_[A-Z0-9_]*(DRAM|MALL|IO|GMI)[A-Za-z0-9_]*|TCC_EA0_(RD|WR)REQ[A-Za-z0-9_]*|TCC_BUBBLE[A-Za-z0-9_]*|MALL[A-Za-z0-9_]*|HBM[A-Za-z0-9_]*|DRAM[A-Za-z0-9_]*)\b" $REPO/$OUT/avail.txt | sort -u > $REPO/$OUT/candidates.txt
echo "candidates: $(tr '\n' ' ' < $REPO/$OUT/candidates.txt)"
i=0
for C in TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM TCC_EA0_WRREQ_DRAM TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum; do
  grep -q -w "$C" $REPO/$OUT/avail.txt || { echo "absent: $C"; continue; }
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE $C -d $REPO/$OUT -o dram$i -- python $REPO/bench.py --steps 1 --warmup 1 --cpu-pairs 0 --also none > $REPO/$OUT/dram$i.log 2>&1
  echo "pass $i ($C) exit $?"
done
cd $REPO
python tools/pmc_summary.py $OUT gemm_ > $OUT/summary.txt 2>&1
rm -f $OUT/*.db $OUT/gemm_traffic.json
grep -c . $OUT/summary.txt
S=$(date +%s)
ONLY="r5: the engine with options vit_fp16 + enc_fp16 + dec_fp16 (what ships at the end of round 5);r5: the engine as shipped in round 5 (precise decoder + vit_fp16 + enc_fp16);r5: decoder floor (vit proj enc exact)"
timeout 600 python tools/error_attribution.py --device cuda --model clip-flant5-xxl --pairs 128 --chunk 32 --only "$ONLY" --out $OUT/attr_xxl_final > $OUT/attr_xxl_final.log 2>&1
echo "[attr $(( $(date +%s) - S )) s, exit $?]"; tail -4 $OUT/attr_xxl_final.log
