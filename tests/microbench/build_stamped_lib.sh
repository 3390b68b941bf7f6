#!/bin/bash
# Builds tests/microbench/ab/libggml-mi355x-log.so: the product library with -DMV_LOG (one in-kernel stamp record per mat-vec launch).
# Use with MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python tests/microbench/frame_stamps.py
set -e
cd "$(dirname "$0")/../../moshi.cpp_amd"
mkdir -p ../tests/microbench/ab
[ -f build/hip_backend.o ] || bash build.sh
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-result -DMV_LOG -c csrc/hip_kernels_fused.hip -o ../tests/microbench/ab/fused_log.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-result -DCH_LOG -c csrc/hip_chain.hip -o ../tests/microbench/ab/chain_log.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../tests/microbench/ab/libggml-mi355x-log.so build/ggml_core.o build/ggml_backend.o build/gguf.o build/hip_backend.o build/hip_kernels_generic.o ../tests/microbench/ab/fused_log.o ../tests/microbench/ab/chain_log.o -Wl,-soname,libggml-mi355x.so
cp libmoshi-hot.so ../tests/microbench/ab/libmoshi-hot.so
echo built
