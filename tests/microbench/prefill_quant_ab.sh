for Q in q8_0 q4_0; do for LIB in tests/microbench/ab/libggml-mi355x-a8d.so moshi.cpp_amd/libggml-mi355x.so; do
echo "== $Q $LIB 4096"; MI355X_LIB=$LIB PREFILL_QUANT=$Q timeout 600 python tests/microbench/prefill_bench.py 64 16,32,64 2>&1 | tail -4
echo "== $Q $LIB 2048"; MI355X_LIB=$LIB PREFILL_QUANT=$Q PREFILL_DIM=2048 timeout 600 python tests/microbench/prefill_bench.py 64 16,32,64 2>&1 | tail -4
done; done
