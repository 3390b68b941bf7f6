#!/bin/bash
# A/B two builds of the library in the reference's own --bench sampling mode (temp 0.8 / 0.7), alternating runs:  ab_bench_sampled.sh <lib_a.so> <lib_b.so> [pairs]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do for L in $A $B; do
  export MI355X_LIB=$L
  timeout 200 python bench.py --sampled --steps 60 --warmup 8 --no-cpu-baseline --no-roofline --no-extras | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.environ['MI355X_LIB'], d['value'], d['phase_us'])"
done; done
