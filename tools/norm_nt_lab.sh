#!/bin/bash
# Lab: libraries whose norm kernels use non-temporal loads / stores of the fp32 stream (build/lab/libvqs_nnt<N>.so)
set -e
cd "$(dirname "$0")/.."
make -C t2v_metrics_amd/csrc > /dev/null
mkdir -p build/lab
for N in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVQS_NORM_NT=$N -c t2v_metrics_amd/csrc/elementwise.hip -o build/lab/elementwise_nnt$N.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lab/libvqs_nnt$N.so build/obj/gemm.hip.o build/obj/attn.hip.o build/lab/elementwise_nnt$N.o build/obj/vqs_api.cpp.o build/obj/vqs_qwen.cpp.o ) &
done
wait
ls -la build/lab | grep nnt
