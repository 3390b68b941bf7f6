#!/bin/bash
# bench-level sweep of two environment tunables: sweep_bench_env.sh VAR1 "v1 v2 .." VAR2 "w1 w2 .."
V1=$1; L1=$2; V2=$3; L2=$4
for a in $L1; do for b in $L2; do
  export $V1=$a $V2=$b
  timeout 200 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('$V1', os.environ['$V1'], '$V2', os.environ['$V2'], d['value'], d['phase_us'])"
done; done
