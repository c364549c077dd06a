#!/bin/bash
# Lab: libraries whose attention kernels are built with extra -D flags: build/lab/libvqs_attn_<tag>.so
# usage: tools/attn_variant_lab.sh tag1 "-DVQS_ATTN_STAGGER=1" tag2 "-DVQS_ATTN_STAGGER=3 -DVQS_ATTN_PRIO=1" ...
set -e
cd "$(dirname "$0")/.."
make -C t2v_metrics_amd/csrc > /dev/null
mkdir -p build/lab
rm -f build/lab/libvqs_attn_*.so
while [ $# -ge 2 ]; do
  TAG=$1; FLAGS=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize $FLAGS -c t2v_metrics_amd/csrc/attn.hip -o build/lab/attn_$TAG.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lab/libvqs_attn_$TAG.so build/obj/gemm.hip.o build/lab/attn_$TAG.o build/obj/elementwise.hip.o build/obj/vqs_api.cpp.o build/obj/vqs_qwen.cpp.o ) &
done
wait
ls build/lab | grep libvqs_attn
