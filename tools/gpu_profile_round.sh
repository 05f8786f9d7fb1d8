#!/bin/bash
# Round profile: bench JSON (with cpu baseline) + rocprofv3 kernel stats of the same command (without cpu leg).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python bench.py > gpurun_out/bench_full.log 2>&1; tail -n 1 gpurun_out/bench_full.log > gpurun_out/bench_line.json; cut -c1-600 gpurun_out/bench_line.json
cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
cd $R
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats.csv
grep '"metric"' gpurun_out/prof.log | tail -1 > gpurun_out/bench_under_rocprof.json
find gpurun_out/prof -name "*kernel_trace.csv" -delete
head -12 gpurun_out/kernel_stats.csv | cut -c1-150
# the same steps with one launch at a time (what bench.py's `roofline.achieved` is measured on): per-kernel averages
# without the other model's workgroups inside them
cd /tmp && export DEVIT_TEACHER_STREAM=0 DEVIT_WGRAD_STREAM=0 && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_serial -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --teacher-lookahead 0 > $R/gpurun_out/prof_serial.log 2>&1
unset DEVIT_TEACHER_STREAM DEVIT_WGRAD_STREAM; cd $R
f=$(find gpurun_out/prof_serial -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats_serial.csv
find gpurun_out/prof_serial -name "*kernel_trace.csv" -delete
grep '"metric"' gpurun_out/prof_serial.log | tail -1 | cut -c1-200
