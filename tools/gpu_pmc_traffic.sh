#!/bin/bash
# HBM-side traffic of every kernel of the bench step: two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE do not
# fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only, python3 directly after `--`.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout 1200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --teacher-lookahead 0 > $R/gpurun_out/pmc_$c.log 2>&1
  tail -n 2 $R/gpurun_out/pmc_$c.log | cut -c1-200
done
cd $R
python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/traffic.json
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" -size +20M -delete
