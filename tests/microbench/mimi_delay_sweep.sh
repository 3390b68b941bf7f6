#!/bin/bash
# serial bench line (frames/s, codec phases) against the Mimi transformer program's first-poll delays (MI355X_MIMI_DELAY="mat-vec phases,out_proj of non-owners")
for d in "$@"; do
  echo -n "delay $d: "; MI355X_MIMI_DELAY=$d python bench.py --serial --steps 60 --warmup 10 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phase_us']['mimi_encode'], d['phase_us']['mimi_decode'])"
done
echo -n "launches: "; MI355X_CHAIN_MIMI=0 python bench.py --serial --steps 60 --warmup 10 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phase_us']['mimi_encode'], d['phase_us']['mimi_decode'])"
