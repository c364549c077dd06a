#!/bin/bash
# Round 5, fourth GPU call: quad tile-order sweep + attention fp16/bf16 + hipBLASLt kernel names; what-if rows on the device
OUT=gpurun_out/r5c4; mkdir -p $OUT; export PYTHONUNBUFFERED=1
REPO=$(pwd)
t() { S=$(date +%s); "$@"; echo "[$(( $(date +%s) - S )) s, exit $?] $*" | cut -c1-200; }
t timeout 600 python tools/lab_gemm_r5.py SA > $OUT/lab_gemm.log 2>&1; grep -v amdgpu $OUT/lab_gemm.log | cut -c1-700
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/$OUT -o hipblaslt -- python $REPO/tools/lab_gemm_r5.py H > $REPO/$OUT/lab_gemm_H.log 2>&1 )
python tools/rocpd_summary.py $OUT/hipblaslt_results.db $OUT/hipblaslt_summary "rocprofv3 --kernel-trace --stats -- python tools/lab_gemm_r5.py H" > /dev/null 2>&1
rm -f $OUT/*.db; grep -i "cijk\|MT[0-9]" $OUT/hipblaslt_summary.md | cut -c1-600 | head -8
ONLY="engine as shipped at the end of round 4 (precise decoder + option vit_fp16);r5: the engine as shipped in round 5 (precise decoder + vit_fp16 + enc_fp16);r5: as round 5 but only the attention sub-block in fp16 (FFN norm output and wi stay bf16);r5: round 5 + decoder cross score path (q, q.Wk, probabilities) and the encoder output in fp16;r5: decoder floor (vit proj enc exact)"
t timeout 600 python tools/error_attribution.py --device cuda --model clip-flant5-xxl --pairs 128 --chunk 32 --only "$ONLY" --out $OUT/attr_xxl_r5 > $OUT/attr_xxl_r5.log 2>&1
grep -v "^#" $OUT/attr_xxl_r5.log | grep -v amdgpu | cut -c1-220 | tail -8
