#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace stats and FETCH_SIZE counters of `python3 bench.py` (eager launches under the profiler:
# rocprofv3 cannot follow hipGraph replays here), condensed into gpurun_out/ for copying to profiles/.
set -x
export TMPDIR=/tmp
R=${1:-r05}
rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_trace -o t --output-format csv -- python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras --backend-flags 2 > gpurun_out/${R}_bench_eager_under_rocprof.json 2> gpurun_out/${R}_trace.err
f=$(find gpurun_out/${R}_trace -name "*kernel_stats.csv" | head -1)
python3 tests/profile_summary.py stats $f gpurun_out/${R}_bench_kernel_stats_eager.csv
t=$(find gpurun_out/${R}_trace -name "*kernel_trace.csv" | head -1)
python3 tests/trace_summary.py $t $(python3 -c "import json; print(json.load(open('gpurun_out/${R}_bench_eager_under_rocprof.json'))['frames_stepped_total'])") > gpurun_out/${R}_bench_kernel_trace_summary.txt 2>&1
rm -rf gpurun_out/${R}_trace
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${R}_pmc -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --backend-flags 2 > gpurun_out/${R}_pmc_bench.json 2> gpurun_out/${R}_pmc.err
c=$(find gpurun_out/${R}_pmc -name "*counter_collection.csv" | head -1)
python3 tests/profile_summary.py pmc $c gpurun_out/${R}_pmc_fetch_size.json
rm -rf gpurun_out/${R}_pmc
python3 bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
# in-kernel timelines of one graph-replayed frame (needs the stamped build: tests/microbench/build_stamped_lib.sh, done in the build container)
if [ -f tests/microbench/ab/libggml-mi355x-log.so ]; then
  MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python3 tests/microbench/frame_stamps.py > gpurun_out/${R}_frame_stamps_in_graph.txt 2>&1
  MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python3 tests/microbench/chain_stamps.py 30 > gpurun_out/${R}_chain_stamps.txt 2>&1
fi
