#!/bin/bash
# tts-shaped bench line (frames/s, Depth us) against the Q8_0 step program's first-poll delays (MI355X_NEST80_DELAY="mat-vec phases,out_proj of non-owners,owners' granule poll")
for d in "$@"; do
  echo -n "delay $d: "; MI355X_NEST80_DELAY=$d python bench.py --model tts_like --steps 40 --warmup 20 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['value_serial'], d['phase_us']['depth'])"
done
