#!/bin/bash
# round 3, first GPU call: the lean-tail tests, then a same-box A/B of the step with the last block full / on its token rows
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_lean.py -x -q -m gpu 2>&1 | tee gpurun_out/r03a_lean_tests.log
for rep in 1 2; do
  DEVIT_LEAN_TAIL=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03a_bench_full_$rep.json 2> gpurun_out/r03a_bench_full_$rep.err
  DEVIT_LEAN_TAIL=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03a_bench_lean_$rep.json 2> gpurun_out/r03a_bench_lean_$rep.err
done
python - <<'PY'
import json
for k in ("full_1","lean_1","full_2","lean_2"):
    d=json.load(open(f"gpurun_out/r03a_bench_{k}.json"))
    print(k, d["value"], d["ms_per_step"], d["host_ms_per_step_idle_queue"], d["roofline"]["frac"])
PY
