#!/bin/bash
# bench-level sweep of the small-matrix mat-vec launch shape (tiles per wave floor, minimum grid)
for tpw in ${TPWS:-1 2 4 8}; do for gm in ${GMS:-128 256 512}; do
  export MI355X_MV_TPW=$tpw MI355X_MV_GRID_MIN=$gm
  timeout 200 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('tpw', os.environ['MI355X_MV_TPW'], 'grid_min', os.environ['MI355X_MV_GRID_MIN'], d['value'], d['phase_us']['temporal'], d['phase_us']['depth'])"
done; done
