#!/bin/bash
# A/B of library builds on the small-batch latency table (tools/bench_latency.py --no-graph): the product library and the lab builds named on
# the command line (lab_so/libvqs_<name>.so), one process each; the digests say whether the scores are the product's bits.
OUT=gpurun_out/${TAG:-small_batch}; mkdir -p $OUT
for L in product "$@"; do
  if [ "$L" = product ]; then unset VQS_LIB_PATH; else export VQS_LIB_PATH=$PWD/lab_so/libvqs_$L.so; fi
  timeout 900 python tools/bench_latency.py --model ${MODEL:-clip-flant5-xxl} --batches ${BATCHES:-1,2,4,8,16,32} --no-graph 2>/dev/null | tee -a $OUT/latency_ab.jsonl
done
