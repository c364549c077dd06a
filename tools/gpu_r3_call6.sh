#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl
L=build/lab/libvqs_quad_abl
timeout 300 python tools/lab_ring2.py quad=t2v_metrics_amd/libvqs_hip.so nodma=${L}.so nulldesc=${L}2.so nowait=${L}4.so ntW=${L}8.so ntA=${L}16.so ntAW=${L}24.so --no-check nodma --no-check nulldesc --no-check nowait --variant 10 > gpurun_out/lab_quad.log 2>&1; echo "quad exit $?"; tail -9 gpurun_out/lab_quad.log | cut -c1-700
