#!/bin/bash
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_cli.py tests/test_gpu_lean.py -x -q -m gpu -k "compacted or shrink or lean" 2>&1 | tail -30 | tee gpurun_out/r03d_tests.log
for mode in masked compact; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --shrink 0.3 --shrink-mode $mode > gpurun_out/r03d_bench_shrink_$mode.json 2> gpurun_out/r03d_bench_shrink_$mode.err
done
python - <<'PY'
import json
for k in ("masked","compact"):
    d=json.load(open(f"gpurun_out/r03d_bench_shrink_{k}.json"))
    print(k, d["value"], d["ms_per_step"], d["host_ms_per_step_idle_queue"], d["config"]["shrink"])
PY
