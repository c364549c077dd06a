#!/bin/bash
# First GPU contact of the ring GEMM form (lab libraries: `make -C t2v_metrics_amd/csrc lab` = four-slot ring, 160 KiB of LDS;
# `make -C t2v_metrics_amd/csrc lab LABDIR=../../build/lab_s3 LABFLAGS=-DVQS_RING_SLOTS=3` = three-slot ring, 128 KiB): bitwise check
# against variant 0 on ten shapes x three tile orders x three repetitions, then its rate next to the 8-wave and wide forms and hipBLASLt.
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl
for L in lab lab_s3; do
  [ -f build/$L/libvqs_hip_lab.so ] || continue
  echo "{\"part\": \"R\", \"library\": \"build/$L/libvqs_hip_lab.so\"}" >> gpurun_out/lab_call.jsonl
  VQS_LIB_PATH=$PWD/build/$L/libvqs_hip_lab.so timeout 200 python tools/lab_call.py --parts R > gpurun_out/lab_ring_$L.log 2>&1
  echo "lab ring ($L) exit $?" >> gpurun_out/lab_ring_$L.log
  tail -18 gpurun_out/lab_ring_$L.log | cut -c1-420
done
