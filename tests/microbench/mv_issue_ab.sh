#!/bin/bash
# round-5 Temporal experiment (i): where the mat-vec's first 2.2 us go. mv_bench (product kernel, -DMV_STAMPS) as built, with the wave's first weight tile
# requested before the activation loads (-DMV_EXP_TILE_FIRST) and with that tile moved by LDS-DMA (-DMV_EXP_TILE0_DMA). Three passes each, same box.
cd "$(dirname "$0")"
for pass in 1 2 3; do
  for b in mv_bench_stamps mv_bench_stamps_tile_first mv_bench_stamps_tile0_dma; do
    echo "== $b (pass $pass)"; ./$b | grep -E "^(out_proj|linear_in|linear_out|text_lin)" | sed 's/ | realtime.*| cyc avg/ | cyc avg/'
  done
done
