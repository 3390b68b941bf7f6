#!/bin/bash
# Rebuilds the host-side objects of libggml-mi355x.so with AddressSanitizer + UBSan (the HIP objects are reused from moshi.cpp_amd/build) and runs
# the CPU test suite against that library. GPU sanitizers are not available on this pool; this covers the C-ABI object model, gguf and the driver.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT=${OUT:-/tmp/asan}
mkdir -p "$OUT"
cd "$ROOT/moshi.cpp_amd"
[ -f build/hip_backend.o ] || bash build.sh
FLAGS="-O1 -g -std=c++17 -fPIC -I../include -Icsrc -ffp-contract=off -fvisibility=hidden -fsanitize=address,undefined -fno-omit-frame-pointer"
for f in ggml_core ggml_backend gguf moshi_hot; do g++ $FLAGS -c csrc/$f.cpp -o "$OUT/$f.o" & done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libggml-mi355x.so" "$OUT"/{ggml_core,ggml_backend,gguf}.o build/hip_backend.o build/hip_kernels_generic.o build/hip_kernels_fused.o \
      -Wl,-soname,libggml-mi355x.so -L"$(dirname "$(gcc -print-file-name=libasan.so)")" -lasan -lubsan
g++ -shared -fPIC -o "$OUT/libmoshi-hot.so" "$OUT/moshi_hot.o" -L"$OUT" -lggml-mi355x -Wl,-soname,libmoshi-hot.so -Wl,-rpath,'$ORIGIN' -L"$(dirname "$(gcc -print-file-name=libasan.so)")" -lasan -lubsan
cd "$ROOT/tests"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  MI355X_LIB="$OUT/libggml-mi355x.so" python -m pytest -x -q -m "not gpu" -p no:cacheprovider
