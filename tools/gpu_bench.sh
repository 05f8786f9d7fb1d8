#!/bin/bash
# bench + rocprof kernel stats.  Output -> gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -n 5 gpurun_out/bench.log
cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
echo "prof exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/prof.log
cd $GRAFT_REPO_ROOT
tail -n 3 gpurun_out/prof.log
find gpurun_out/prof -name "*kernel_stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f"
# keep only the stats csv (the trace is large)
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
