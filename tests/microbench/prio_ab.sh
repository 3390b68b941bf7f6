#!/bin/bash
# stream priorities on / off for the two-stream frame loop
for i in 1 2 3; do for P in 0 1; do
  MI355X_STREAM_PRIORITY=$P timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline --no-extras | P=$P python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('priority', os.environ['P'], d['value'], d.get('serial_loop'))"
done; done
