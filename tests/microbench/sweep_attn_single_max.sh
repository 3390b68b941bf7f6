#!/bin/bash
# round 5: up to how many live slots should a head's first workgroup do the whole attention alone (MI355X_ATTN_SINGLE_MAX, default ATTN_SINGLE_MAX) before the
# head is split over workgroups? bench.py from several ring fills, Temporal us per frame.
for f in 180 260 350 500 800; do
  for sm in 160 256 384 512 768; do
    echo -n "fill $f single_max $sm: "
    MI355X_ATTN_SINGLE_MAX=$sm python bench.py --context-fill $f --steps 12 --warmup 4 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phase_us']['temporal'])"
  done
done
