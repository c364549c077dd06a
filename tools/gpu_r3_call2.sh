#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl
timeout 300 python tools/lab_ring2.py sched1=build/lab/libvqs_hip_lab.so sched0=build/lab_s0/libvqs_hip_lab.so nodma=build/lab_abl/libvqs_hip_lab.so --no-check nodma > gpurun_out/lab_ring2.log 2>&1; echo "ring2 exit $?"; tail -12 gpurun_out/lab_ring2.log | cut -c1-500
VQS_LIB_PATH=$PWD/build/lab/libvqs_hip_lab.so timeout 150 python tools/lab_call.py --parts RT > gpurun_out/lab_rt.log 2>&1; echo "RT exit $?"; tail -7 gpurun_out/lab_rt.log | cut -c1-400
