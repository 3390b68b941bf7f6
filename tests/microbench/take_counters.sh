#!/bin/bash
# Run on the GPU box from the repo root (cd /tmp; export TMPDIR=/tmp first): SQ counters of `python3 bench.py` (eager), three separate --pmc passes, condensed
# into gpurun_out/<round>_sq_counters.txt for copying to profiles/.
set -x
export TMPDIR=/tmp
R=${1:-r05}
EXTRA=${2:-}   # e.g. --serial: the codec graphs then run on the LM stream, the Mimi transformer programs (mimi_tr_kernel) with them
i=0
for set in "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_LDS"; do
  i=$((i + 1))
  rocprofv3 --pmc $set -d gpurun_out/${R}_sq$i -o p --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --backend-flags 2 $EXTRA > gpurun_out/${R}_sq${i}_bench.json 2> gpurun_out/${R}_sq$i.err
done
python3 tests/profile_summary.py counters gpurun_out/${R}_sq_counters.txt $(find gpurun_out/${R}_sq1 gpurun_out/${R}_sq2 gpurun_out/${R}_sq3 -name "*counter_collection.csv")
rm -rf gpurun_out/${R}_sq1 gpurun_out/${R}_sq2 gpurun_out/${R}_sq3
