#!/bin/bash
# sweep one env tunable at several context fills: sweep_ctx_env.sh VAR "v1 v2 .." "fill1 fill2 .."
V=$1; L=$2; F=$3
for cf in $F; do for a in $L; do
  export $V=$a
  timeout 300 python bench.py --steps 60 --context-fill $cf --no-cpu-baseline --no-roofline | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('$V', os.environ['$V'], 'fill', d['config']['context_fill_start'], d['value'], d['phase_us']['temporal'])"
done; done
