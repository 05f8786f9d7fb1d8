#!/bin/bash
# attention backward: 4-wave two-per-CU kernel vs the 8-wave kernel (DEVIT_ATTN_BWD=8): correctness, cold timings, step A/B
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_lean.py -x -q -m gpu -k "attention or attn or lean" 2>&1 | tail -8 | tee gpurun_out/r03i_tests.log
for rep in 1 2; do
  DEVIT_ATTN_BWD=8 timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/bwd8 /' | tee -a gpurun_out/r03i_attn_ab.txt
  DEVIT_ATTN_BWD=4 timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/bwd4 /' | tee -a gpurun_out/r03i_attn_ab.txt
done
for rep in 1 2; do
  for v in 8 4; do
    DEVIT_ATTN_BWD=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03i_bench_bwd${v}_$rep.json 2> gpurun_out/r03i_bench.err
  done
done
python - <<'PY' | tee -a gpurun_out/r03i_attn_ab.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r03i_bench_*.json")):
    d=json.load(open(f))
    print(f.split("r03i_bench_")[1].ljust(14), d["value"], "img/s", d["ms_per_step"], "ms", d["roofline"]["hbm_bound_kernels"].get("attention_bwd"))
PY
