#!/bin/bash
# First GPU contact of the ring GEMM form (lab library, build/lab/libvqs_hip_lab.so = `make -C t2v_metrics_amd/csrc lab`): bitwise check
# against variant 0 on ten shapes x three tile orders x three repetitions, then its rate next to the 8-wave and wide forms and hipBLASLt.
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl
VQS_LIB_PATH=$PWD/build/lab/libvqs_hip_lab.so timeout 300 python tools/lab_call.py --parts R > gpurun_out/lab_ring.log 2>&1
echo "lab ring exit $?" >> gpurun_out/lab_ring.log
tail -20 gpurun_out/lab_ring.log | cut -c1-500
