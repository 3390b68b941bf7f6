#!/bin/bash
# alternate several builds: ab_bench3.sh <rounds> <lib> <lib> ...
N=$1; shift
for i in $(seq $N); do for L in "$@"; do
  export MI355X_LIB=$L
  timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.path.basename(os.environ['MI355X_LIB']), d['value'], d['phase_us'])"
done; done
