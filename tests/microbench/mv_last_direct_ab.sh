#!/bin/bash
# round-5 Temporal experiment (ii): a wave's last tile in the 8-lanes-per-super-block form (-DMV_EXP_LAST_DIRECT) against the product kernel, mv_bench with stamps
cd "$(dirname "$0")"
for pass in 1 2 3; do
  for b in mv_bench_stamps mv_bench_stamps_last_direct; do
    echo "== $b (pass $pass)"; ./$b | grep -E "(in_proj|out_proj|linear_in|linear_out|text_lin) " | sed 's/ | realtime.*| cyc avg/ | cyc avg/'
  done
done
