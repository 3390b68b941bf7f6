for sh in 126 60 124 62; do echo "=== shapes mask $sh"; MI355X_CHAIN_SHAPES=$sh MI355X_CHAIN=1 MI355X_CHAIN_GRID=256 MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so timeout 300 python tests/microbench/chain_stamps.py 0 2>&1 | grep "descriptor reads\|mean per"; done
echo "--- unchained"; python tests/microbench/lm_only.py 40 16
for sh in 126 60 124 62; do echo "--- grid 256 shapes mask $sh"; MI355X_CHAIN_SHAPES=$sh MI355X_CHAIN_GRID=256 python tests/microbench/lm_only.py 40 32; done
