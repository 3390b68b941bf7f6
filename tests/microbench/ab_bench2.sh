#!/bin/bash
# A/B two builds with extra bench arguments: ab_bench2.sh <lib_a> <lib_b> <pairs> <bench args...>
A=$1; B=$2; N=$3; shift 3
for i in $(seq $N); do for L in $A $B; do
  export MI355X_LIB=$L
  timeout 300 python bench.py --no-cpu-baseline --no-roofline "$@" | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.path.basename(os.environ['MI355X_LIB']), d['steps'], d['config']['context_fill_start'], d['value'], d['phase_us']['temporal'])"
done; done
