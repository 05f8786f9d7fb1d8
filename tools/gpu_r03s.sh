#!/bin/bash
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_lean.py -x -q -m gpu -k "attention or attn or lean" 2>&1 | tail -3 | tee gpurun_out/r03s_tests.log
DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_attnstamp.so timeout -k 5 120 python tools/attn_fwd_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03s_attn_fwd_stamps.txt
rm -f gpurun_out/r03s_attn_ab.txt
for rep in 1 2 3; do
  DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_fwd8.so timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/fwd8 /' | tee -a gpurun_out/r03s_attn_ab.txt
  timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/fwd4 /' | tee -a gpurun_out/r03s_attn_ab.txt
done
for rep in 1 2; do
  DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_fwd8.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03s_bench_fwd8_$rep.json 2> gpurun_out/r03s_bench.err
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03s_bench_fwd4_$rep.json 2> gpurun_out/r03s_bench.err
done
python - <<'PY' | tee -a gpurun_out/r03s_attn_ab.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r03s_bench_*.json")):
    d=json.load(open(f))
    print(f.split("r03s_bench_")[1].ljust(14), d["value"], "img/s", d["ms_per_step"], "ms", d["roofline"]["hbm_bound_kernels"].get("attention_fwd"))
PY
