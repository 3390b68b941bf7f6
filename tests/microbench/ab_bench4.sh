#!/bin/bash
# alternate several builds with the codec rings full (250 slots after 250 frames): ab_bench4.sh <rounds> <lib> <lib> ...
N=$1; shift
for i in $(seq $N); do for L in "$@"; do
  export MI355X_LIB=$L
  timeout 300 python bench.py --steps 100 --warmup 260 --no-cpu-baseline --no-roofline --no-extras | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.path.basename(os.environ['MI355X_LIB']), d['value'], d['phase_us'])"
done; done
