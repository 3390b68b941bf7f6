#!/bin/bash
# alternate builds at several context fills: ab_ctx.sh <rounds> <lib> <lib> ...   (FILLS="300 1000 2800")
N=$1; shift
for F in ${FILLS:-300 1000 2800}; do for i in $(seq $N); do for L in "$@"; do
  export MI355X_LIB=$L
  timeout 300 python bench.py --steps 40 --warmup 6 --context-fill $F --no-cpu-baseline --no-roofline --no-extras ${BENCH_ARGS} | F=$F python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('fill', os.environ['F'], os.path.basename(os.environ['MI355X_LIB']), d['value'], d['phase_us']['temporal'])"
done; done; done
