#!/bin/bash
# Round 3, GPU call 1: first contact of the ring GEMM form (bitwise vs variant 0, rate), its cycle split, and the attention
# kernel's per-phase s_memtime split.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl
bash tools/gpu_lab_ring.sh
if [ -f build/lab_timing/libvqs_hip_lab.so ]; then
  VQS_LIB_PATH=$PWD/build/lab/libvqs_hip_lab.so timeout 150 python tools/lab_call.py --parts RT > gpurun_out/lab_rt.log 2>&1; echo "RT exit $?"; tail -8 gpurun_out/lab_rt.log | cut -c1-400
fi
timeout 150 python tools/lab_call.py --parts AT > gpurun_out/lab_at.log 2>&1; echo "AT exit $?"; tail -6 gpurun_out/lab_at.log | cut -c1-600
