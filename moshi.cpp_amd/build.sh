#!/bin/bash
# Builds the MI355X ggml drop-in (libggml-mi355x.so: the boundary library, nothing but the ggml C API) and, next to it, the test / bench
# harness (libmoshi-hot.so: synthetic weights + the frame driver of include/moshi_hot.h, linked AGAINST the drop-in through its public
# surface only) for gfx950. hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
SRC=csrc
OUT=${OUT:-.}
mkdir -p build
ARCH=${ARCH:-gfx950}
CXXFLAGS="-O3 -std=c++17 -fPIC -I../include -I$SRC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function"
HIPFLAGS="--offload-arch=$ARCH $CXXFLAGS -Wno-unused-result"
pids=()
for f in ggml_core ggml_backend gguf moshi_hot; do
  [ -f $SRC/$f.cpp ] || continue
  if [ ! -f build/$f.o ] || [ $SRC/$f.cpp -nt build/$f.o ] || [ -n "$(find $SRC ../include -name '*.h' -newer build/$f.o)" ]; then
    g++ $CXXFLAGS -c $SRC/$f.cpp -o build/$f.o & pids+=($!)
  fi
done
for f in hip_backend hip_kernels_generic hip_kernels_fused hip_chain; do
  if [ ! -f build/$f.o ] || [ $SRC/$f.hip -nt build/$f.o ] || [ -n "$(find $SRC ../include -name '*.h' -newer build/$f.o)" ]; then
    hipcc $HIPFLAGS -c $SRC/$f.hip -o build/$f.o & pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=$ARCH -shared -fPIC -o $OUT/libggml-mi355x.so build/ggml_core.o build/ggml_backend.o build/gguf.o build/hip_backend.o build/hip_kernels_generic.o build/hip_kernels_fused.o build/hip_chain.o -Wl,-soname,libggml-mi355x.so
g++ -shared -fPIC -o $OUT/libmoshi-hot.so build/moshi_hot.o -L$OUT -lggml-mi355x -Wl,-soname,libmoshi-hot.so -Wl,-rpath,'$ORIGIN'
echo "built $OUT/libggml-mi355x.so $OUT/libmoshi-hot.so"
