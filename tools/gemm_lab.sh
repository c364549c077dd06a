#!/bin/bash
# Build ablated copies of the library (lab only): build/lab/libvqs_abl<N>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build/lab
for A in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVQS_ABLATE=$A -shared -o build/lab/libvqs_abl$A.so \
     t2v_metrics_amd/csrc/gemm.hip t2v_metrics_amd/csrc/attn.hip t2v_metrics_amd/csrc/elementwise.hip t2v_metrics_amd/csrc/vqs_api.cpp &
done
wait
ls -la build/lab
