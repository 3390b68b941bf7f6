#!/bin/bash
# codec-only measurements on the GPU box: plan listing, replayed us/frame (program on / off), eager kernel stats under rocprofv3, Mimi program stamps
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
MI355X_DUMP_PLAN=1 python tests/microbench/codec_plan_dump.py 32 2> gpurun_out/r05_codec_plan.txt >/dev/null
for w in enc dec; do
  python tests/microbench/mimi_only.py $w 300 32 > gpurun_out/r05_codec_${w}_replay.txt 2>&1
  MI355X_CHAIN_MIMI=0 python tests/microbench/mimi_only.py $w 300 32 >> gpurun_out/r05_codec_${w}_replay.txt 2>&1
  MI355X_CONV_SCATTER=0 python tests/microbench/mimi_only.py $w 300 32 >> gpurun_out/r05_codec_${w}_replay.txt 2>&1
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o p --output-format csv -- python3 tests/microbench/mimi_only.py $w 40 34 > /tmp/prof_$w.log 2>&1
  f=$(find /tmp/prof_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/r05_codec_${w}_kernel_stats.csv || tail -5 /tmp/prof_$w.log
  f=$(find /tmp/prof_$w -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/r05_codec_${w}_kernel_trace.csv
done
if [ -f tests/microbench/ab/libggml-mi355x-log.so ]; then
MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python tests/microbench/mimi_stamps.py > gpurun_out/r05_mimi_stamps.txt 2>&1
fi
cat gpurun_out/r05_codec_enc_replay.txt gpurun_out/r05_codec_dec_replay.txt gpurun_out/r05_mimi_stamps.txt
