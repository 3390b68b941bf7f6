for tpw in 4 8 12 16 24; do for gm in 1 200; do
  echo "TPW=$tpw GRID_MIN=$gm:"; MI355X_MV_TPW=$tpw MI355X_MV_GRID_MIN=$gm timeout 60 tests/microbench/mv_bench 2>&1 | cut -c1-75
done; done
