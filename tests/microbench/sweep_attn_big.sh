#!/bin/bash
# split-attention range doubling threshold (MI355X_ATTN_BIG_MIN) and NPRE-independent knobs at long context
for F in 1000 2000 2800; do for B in 1024 100000; do for S in 128; do
  MI355X_ATTN_BIG_MIN=$B MI355X_ATTN_SLOTS=$S timeout 300 python bench.py --steps 40 --warmup 6 --context-fill $F --no-cpu-baseline --no-roofline --no-extras | F=$F B=$B S=$S python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('fill', os.environ['F'], 'big_min', os.environ['B'], 'slots', os.environ['S'], d['value'], d['phase_us']['temporal'])"
done; done; done
