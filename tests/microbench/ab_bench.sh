#!/bin/bash
# A/B two builds of the library at bench level, alternating runs (GPU clocks drift by a few per cent between runs):
#   ab_bench.sh <lib_a.so> <lib_b.so> [pairs]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do for L in $A $B; do
  export MI355X_LIB=$L
  timeout 200 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.path.basename(os.environ['MI355X_LIB']), d['value'], d['phase_us'])"
done; done
