#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl
L=build/lab/libvqs_quad_abl
timeout 300 python tools/lab_ring2.py quad=t2v_metrics_amd/libvqs_hip.so mfma16=${L}32.so --no-check mfma16 --variant 10 > gpurun_out/lab_quad.log 2>&1; echo "quad exit $?"; tail -8 gpurun_out/lab_quad.log | cut -c1-500
