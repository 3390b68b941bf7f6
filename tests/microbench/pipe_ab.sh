#!/bin/bash
# serial vs software-pipelined frame loop, alternated
for i in 1 2 3; do for M in "--serial" ""; do
  timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline --no-extras $M | M="$M" python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.environ['M'] or 'pipelined', d['value'], d.get('serial_loop'), d['phase_us'])"
done; done
