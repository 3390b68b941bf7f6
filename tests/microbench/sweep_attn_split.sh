# split-attention geometry at long context after the granule hand-off: range size (MI355X_ATTN_SLOTS) and the live length beyond which ranges double (MI355X_ATTN_BIG_MIN)
for cfg in "128 1024" "128 4096" "64 4096" "64 1024" "256 4096"; do set -- $cfg
  for fill in 2800 600; do
  MI355X_ATTN_SLOTS=$1 MI355X_ATTN_BIG_MIN=$2 python bench.py --context-fill $fill --steps 50 --warmup 8 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slots $1 big_min $2 fill $fill:', d['value'], d['phase_us']['temporal'])"
  done
done
