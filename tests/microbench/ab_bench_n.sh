#!/bin/bash
# A/B several builds of the library at bench level, round-robin (GPU clocks drift between runs):  ab_bench_n.sh <pairs> <lib.so> [<lib.so> ...]
N=$1; shift
for i in $(seq $N); do for L in "$@"; do
  export MI355X_LIB=$L
  timeout 200 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline --no-extras | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.environ['MI355X_LIB'].split('/')[-2], d['value'], d['phase_us']['temporal'])"
done; done
