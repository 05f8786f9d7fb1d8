#!/bin/bash
# experiment: attention forward moves whole 128-byte rows per memory instruction: output through a per-wave LDS slab (DEVIT_ATTN_OUT_ROWS), Q fragments by a lane trade (DEVIT_ATTN_Q_ROWS)
set -eo pipefail
mkdir -p gpurun_out; rm -f gpurun_out/r03B_attn_ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lean.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r03B_tests.txt 2>&1 || { tail -30 gpurun_out/r03B_tests.txt; exit 1; }
tail -2 gpurun_out/r03B_tests.txt
for rep in 1 2 3; do
  timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/out + Q  /' | tee -a gpurun_out/r03B_attn_ab.txt
  DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_qrows0.so timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/out only /' | tee -a gpurun_out/r03B_attn_ab.txt
  DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_orows0.so timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/neither  /' | tee -a gpurun_out/r03B_attn_ab.txt
done
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03B_bench_rows_$rep.json 2> gpurun_out/r03B_bench.err
  DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_orows0.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03B_bench_old_$rep.json 2> gpurun_out/r03B_bench.err
done
python - <<'PY' | tee -a gpurun_out/r03B_attn_ab.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r03B_bench_*.json")):
    d=json.load(open(f)); h=d["roofline"]["hbm_bound_kernels"]
    print(f.split("r03B_bench_")[1].ljust(16), d["value"], "img/s", d["ms_per_step"], "ms fwd", h["attention_fwd"]["ms_per_step"], "bwd", h["attention_bwd"]["ms_per_step"])
PY
