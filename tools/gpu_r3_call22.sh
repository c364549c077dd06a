#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl
L=build/lab/libvqs_quad_abl
timeout 300 python tools/lab_ring2.py quad16=t2v_metrics_amd/libvqs_hip.so noepi=${L}128.so nostores=${L}256.so --no-check noepi --no-check nostores --variant 3 --tol > gpurun_out/lab_quad.log 2>&1; echo "quad exit $?"; tail -8 gpurun_out/lab_quad.log | cut -c1-420 | grep -v stats
