#!/bin/bash
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "ragged or compacted or gemm" 2>&1 | tail -3 | tee gpurun_out/r03n_tests.log
for rep in 1 2; do for mode in masked compact; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --shrink 0.3 --shrink-mode $mode > gpurun_out/r03n_bench_shrink_${mode}_$rep.json 2> gpurun_out/r03n_bench.err
done; done
python - <<'PY' | tee gpurun_out/r03n_summary.txt
import json
for rep in (1,2):
  for k in ("masked","compact"):
    d=json.load(open(f"gpurun_out/r03n_bench_shrink_{k}_{rep}.json"))
    print(k, rep, d["value"], "img/s", d["ms_per_step"], "ms", d["config"]["shrink"])
PY
