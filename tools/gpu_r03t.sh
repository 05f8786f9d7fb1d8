#!/bin/bash
set -eo pipefail
mkdir -p gpurun_out; rm -f gpurun_out/r03t_attn_ab.txt
for rep in 1 2 3; do
  timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/default /' | tee -a gpurun_out/r03t_attn_ab.txt
  DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_attnnt.so timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/nt      /' | tee -a gpurun_out/r03t_attn_ab.txt
done
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03t_bench_default_$rep.json 2> gpurun_out/r03t_bench.err
  DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_attnnt.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03t_bench_nt_$rep.json 2> gpurun_out/r03t_bench.err
done
python - <<'PY' | tee -a gpurun_out/r03t_attn_ab.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r03t_bench_*.json")):
    d=json.load(open(f)); h=d["roofline"]["hbm_bound_kernels"]
    print(f.split("r03t_bench_")[1].ljust(16), d["value"], "img/s", d["ms_per_step"], "ms fwd", h["attention_fwd"]["ms_per_step"], "bwd", h["attention_bwd"]["ms_per_step"])
PY
