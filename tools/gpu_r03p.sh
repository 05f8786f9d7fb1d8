#!/bin/bash
# experiment: balanced persistent grids (every workgroup the same number of tiles, fewer workgroups)
set -eo pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do for b in 0 1 2; do
  DEVIT_GEMM_BALANCE=$b timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03p_bench_balance${b}_$rep.json 2> gpurun_out/r03p_bench.err
done; done
python - <<'PY' | tee gpurun_out/r03p_summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r03p_bench_*.json")):
    d=json.load(open(f)); r=d["roofline"]
    print(f.split("r03p_bench_")[1].ljust(18), d["value"], "img/s", d["ms_per_step"], "ms | dominant template", r["achieved"], "TF/s serial,", r["in_two_stream_timed_region"]["achieved"], "two-stream | gemm ms", r["gemm_ms_per_step"])
PY
