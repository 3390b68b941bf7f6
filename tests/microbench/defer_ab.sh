#!/bin/bash
# decode half of the codec beside the Temporal graph (default) vs deferred to start when it has finished (MOSHI_HOT_DEFER_DECODE=1)
for i in 1 2 3; do for D in 0 1; do
  if [ $D = 1 ]; then export MOSHI_HOT_DEFER_DECODE=1; else unset MOSHI_HOT_DEFER_DECODE; fi
  timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline --no-extras | D=$D python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('defer_decode', os.environ['D'], d['value'])"
done; done
