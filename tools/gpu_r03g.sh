#!/bin/bash
# per-kernel stats of the current tree: serialized (one launch at a time) and two-stream
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && export DEVIT_TEACHER_STREAM=0 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_serial -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --teacher-lookahead 0 > $R/gpurun_out/prof_serial.log 2>&1
unset DEVIT_TEACHER_STREAM; cd $R
f=$(find gpurun_out/prof_serial -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03g_kernel_stats_serial.csv
find gpurun_out/prof_serial -name "*kernel_trace.csv" -delete
grep '"metric"' gpurun_out/prof_serial.log | tail -1 | cut -c1-300
head -40 gpurun_out/r03g_kernel_stats_serial.csv | cut -c1-200
