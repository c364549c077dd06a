#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl
timeout 300 python tools/lab_ring2.py quad16=t2v_metrics_amd/libvqs_hip.so stagger1=build/lab/libvqs_quad_stagger1.so stagger2=build/lab/libvqs_quad_stagger2.so --no-check stagger1 --no-check stagger2 --variant 3 --tol --all > gpurun_out/lab_quad.log 2>&1; echo "quad exit $?"; tail -14 gpurun_out/lab_quad.log | cut -c1-420 | grep -v stats
