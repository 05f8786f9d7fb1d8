#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
cd $R
grep '"metric"' gpurun_out/prof.log | tail -1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (8 steps):", tot / 8e6)
for r in rows[:24]:
    print("%-70s calls %5s  avg %9.1f us  per-step %7.3f ms  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 8e6, float(r["Percentage"])))
PY
find gpurun_out/prof -name "*kernel_trace.csv" -delete
