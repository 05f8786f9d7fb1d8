#!/bin/bash
# experiment: teacher forward and student step on disjoint CU partitions (CU-masked streams)
set -eo pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for t in 0 64 96 112 128; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --partition $t > gpurun_out/r03h_bench_part${t}_$rep.json 2> gpurun_out/r03h_bench_part${t}.err || { tail -5 gpurun_out/r03h_bench_part${t}.err; exit 1; }
done; done
python - <<'PY' | tee gpurun_out/r03h_summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r03h_bench_*.json")):
    d=json.load(open(f))
    print(f.split("r03h_bench_")[1].ljust(18), d["value"], "img/s", d["ms_per_step"], "ms | teacher CUs", d["partition_teacher_cus"])
PY
