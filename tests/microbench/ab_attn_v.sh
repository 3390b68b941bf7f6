#!/bin/bash
# A/B of the split attention's V request point (ATTN_V_EARLY): default library vs a build with the other -DATTN_V_EARLY value as tests/microbench/ab/libggml-mi355x-vlate.so, alternating
for i in 1 2 3; do
  for v in early late; do
    if [ $v = late ]; then export MI355X_LIB=tests/microbench/ab/libggml-mi355x-vlate.so; else unset MI355X_LIB; fi
    a=$(python tests/microbench/lm_only.py 100 0 | awk '{print $3}')
    b=$(python bench.py --context-fill 2800 --steps 60 --warmup 8 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['phase_us']['temporal'])")
    echo "$v: lm us/frame (100 frames from empty) $a | fill 2800: $b"
  done
done
