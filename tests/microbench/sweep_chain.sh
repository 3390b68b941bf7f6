timeout 600 python -m pytest tests/test_chain_engine.py -x -q 2>&1 | tail -5
MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so timeout 300 python tests/microbench/chain_stamps.py 0 2>&1 | tail -30
echo "--- unchained"; python tests/microbench/lm_only.py 40 16
for g in 64 96 128 192 256; do echo "--- grid $g"; MI355X_CHAIN_GRID=$g python tests/microbench/lm_only.py 40; done
