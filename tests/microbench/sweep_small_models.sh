#!/bin/bash
# launch-shape sweep of the block mat-vec at the 2048-wide (tts / stt) shapes: frames/s and Temporal microseconds per setting
for model in tts_like stt_like; do
  for bt in 1536 1024 700 512 256; do
    for gm in 128 256; do
      export MI355X_MV_BIG_TILES=$bt MI355X_MV_GRID_MIN=$gm
      timeout 200 python bench.py --model $model --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$model big_tiles=$bt grid_min=$gm', d['value'], d['phase_us']['temporal'], d['phase_us']['depth'])"
    done
  done
done
