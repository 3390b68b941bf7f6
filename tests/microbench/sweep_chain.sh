timeout 600 python -m pytest tests/test_chain_engine.py -x -q 2>&1 | tail -5
MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so timeout 300 python tests/microbench/chain_stamps.py 0 4 2>&1 | tail -36
echo "--- unchained"; python tests/microbench/lm_only.py 40 16
for g in 64 128 256; do for th in "24 16"; do set -- $th; echo "--- grid $g quota $1 dots $2"; MI355X_CHAIN_GRID=$g MI355X_CHAIN_THROTTLE=$1 MI355X_CHAIN_QUOTA_DOTS=$2 python tests/microbench/lm_only.py 40; done; done
