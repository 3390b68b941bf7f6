#!/bin/bash
# The -DMV_EXP_* forms are NOT in the product kernel's source any more (round 6): they live in mv_exp_forms.patch (made against hip_kernels_fused.hip at
# round 5's final commit 80f7b65). To rebuild the variants: `git show 80f7b65:moshi.cpp_amd/csrc/hip_kernels_fused.hip > /tmp/f.hip` (the file WITH the forms), point
# mv_bench.hip's #include at it and build with -DMV_STAMPS -DMV_EXP_<FORM>; or `patch -p0 < mv_exp_forms.patch` on a checkout of that commit's successor.
# round-5 Temporal experiment (i): where the mat-vec's first 2.2 us go. mv_bench (product kernel, -DMV_STAMPS) as built, with the wave's first weight tile
# requested before the activation loads (-DMV_EXP_TILE_FIRST) and with that tile moved by LDS-DMA (-DMV_EXP_TILE0_DMA). Three passes each, same box.
cd "$(dirname "$0")"
for pass in 1 2 3; do
  for b in mv_bench_stamps mv_bench_stamps_tile_first mv_bench_stamps_tile0_dma; do
    echo "== $b (pass $pass)"; ./$b | grep -E "^(out_proj|linear_in|linear_out|text_lin)" | sed 's/ | realtime.*| cyc avg/ | cyc avg/'
  done
done
