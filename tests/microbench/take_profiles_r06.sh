#!/bin/bash
# Round-6 profile set, run on the GPU box from the repo root (tests/microbench/build_stamped_lib.sh first, in the build container). Pass 1 writes the PMC file;
# copy it to profiles/ and run pass 2 (bench lines whose `roofline.traffic` is bound to these sources' hash).
set -x
export TMPDIR=/tmp
R=r06
O=gpurun_out/$R
mkdir -p $O
if [ "$1" != "pass2" ]; then
  bash tests/microbench/take_profiles.sh $R
  mv gpurun_out/${R}_* $O/ 2>/dev/null
  # the long-context line's own profile: the same command from a ring holding 2 800 of 3 000 slots, eager under rocprofv3
  rocprofv3 --kernel-trace --stats -d /tmp/${R}_fill -o t --output-format csv -- python3 bench.py --context-fill 2800 --steps 20 --warmup 4 --no-cpu-baseline --no-extras --backend-flags 2 > $O/${R}_bench_fill2800_eager_under_rocprof.json 2> /tmp/${R}_fill.err
  f=$(find /tmp/${R}_fill -name "*kernel_stats.csv" | head -1); python3 tests/profile_summary.py stats $f $O/${R}_fill2800_kernel_stats_eager.csv
  rm -rf /tmp/${R}_fill
  MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so FILL=2800 python3 tests/microbench/frame_stamps.py > $O/${R}_frame_stamps_fill_2800.txt 2>&1
  MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python3 tests/microbench/nest_sampler_stamps.py > $O/${R}_sampler_phase_stamps.txt 2>&1
  MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python3 tests/microbench/mimi_stamps.py > $O/${R}_mimi_stamps.txt 2>&1
  for w in enc dec; do
    python tests/microbench/mimi_only.py $w 300 32 > $O/${R}_codec_${w}_replay.txt 2>&1
    MI355X_VQ_CHAIN=0 python tests/microbench/mimi_only.py $w 300 32 >> $O/${R}_codec_${w}_replay.txt 2>&1
  done
else
  python3 bench.py > $O/${R}_bench_default.json 2> $O/${R}_bench_default.err
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${R}_bench_driver_command.json 2> /dev/null
  python3 bench.py --serial --no-extras > $O/${R}_bench_serial.json 2> /dev/null
  python3 bench.py --sampled --no-extras > $O/${R}_bench_sampled.json 2> /dev/null
  python3 bench.py --model personaplex --context 2000 --context-fill 1900 --no-extras --no-cpu-baseline > $O/${R}_bench_personaplex_ctx2000.json 2> /dev/null
  python3 -m pytest tests -x -q -m gpu > $O/${R}_gpu_suite.txt 2>&1; tail -3 $O/${R}_gpu_suite.txt
fi
