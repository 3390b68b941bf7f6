#!/bin/bash
# LM-only frame time (Temporal + Depth) against the step program's poll delays (MI355X_NEST_DELAY, s_sleep units per phase kind:
# in_proj, out_proj, linear_in, linear_out, linears[k])
for d in "$@"; do
  echo -n "delay $d: "; MI355X_NEST_DELAY=$d python tests/microbench/lm_only.py 100 | tail -1
done
