#!/bin/bash
# round-5 Temporal experiment (ii), the cheap half: out_proj (one tile per wave: its only tile IS its last) in the 8-lanes-per-super-block form
# (MI355X_MVD_MAX_TILES lifts the product's 512-tile bound) against the LDS-tile form, mv_bench with stamps, cold weights.
cd "$(dirname "$0")"
for pass in 1 2; do
  echo "== LDS tiles (product) pass $pass"; ./mv_bench_stamps | grep -E "^(out_proj)" | sed 's/ | realtime.*| cyc avg/ | cyc avg/'
  for p in 1 2 3 4; do echo "== direct form, $p passes, pass $pass"; MI355X_MVD_MAX_TILES=2048 MI355X_MVD_PASSES=$p ./mv_bench_stamps | grep -E "^(out_proj)" | sed 's/ | realtime.*| cyc avg/ | cyc avg/'; done
done
