#!/bin/bash
# sweep two env tunables at several context fills: sweep_ctx_env2.sh VAR1 "v.." VAR2 "w.." "fills"
V1=$1; L1=$2; V2=$3; L2=$4; F=$5
for cf in $F; do for a in $L1; do for b in $L2; do
  export $V1=$a $V2=$b
  timeout 300 python bench.py --steps 60 --context-fill $cf --no-cpu-baseline --no-roofline | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('$V1', os.environ['$V1'], '$V2', os.environ['$V2'], 'fill', d['config']['context_fill_start'], d['value'], d['phase_us']['temporal'])"
done; done; done
