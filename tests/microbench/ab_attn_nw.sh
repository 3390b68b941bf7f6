#!/bin/bash
# A/B of the split attention's workgroup width (MI355X_ATTN_SPLIT_NW = 4: ranges of 128 / 256 slots, 8: 192 / 384 slots), alternating
for i in 1 2 3; do
  for v in 4 8; do
    export MI355X_ATTN_SPLIT_NW=$v
    a=$(python tests/microbench/lm_only.py 100 0 | awk '{print $3}')
    b=$(python bench.py --context-fill 2800 --steps 60 --warmup 8 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['phase_us']['temporal'])")
    c=$(python bench.py --context-fill 600 --steps 60 --warmup 8 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['phase_us']['temporal'])")
    echo "waves $v: lm us/frame (100 frames from empty) $a | fill 2800: $b | fill 600: $c"
  done
done
