#!/bin/bash
# experiment: GELU / dGELU epilogue arithmetic as packed fp32 instructions + DPP without an old value -- correctness, then A/B against the previous build
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_lean.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r03D_tests.txt 2>&1 || { tail -30 gpurun_out/r03D_tests.txt; exit 1; }
tail -3 gpurun_out/r03D_tests.txt
timeout -k 10 300 python tools/gemm_race_screen.py 12 > gpurun_out/r03D_race.txt 2>&1 || { tail -20 gpurun_out/r03D_race.txt; exit 1; }
tail -2 gpurun_out/r03D_race.txt
for rep in 1 2; do for v in prev main; do
  if [ $v = main ]; then unset DEVIT_LIB_PATH; else export DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so; fi
  COLD=1 timeout -k 10 300 python tools/gemm_bench.py > gpurun_out/r03D_gemm_${v}_$rep.txt 2>&1
done; done
unset DEVIT_LIB_PATH
for rep in 1 2 3; do for v in prev main; do
  if [ $v = main ]; then unset DEVIT_LIB_PATH; else export DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03D_bench_${v}_$rep.json 2> gpurun_out/r03D_bench.err
done; done
unset DEVIT_LIB_PATH
python - <<'PY' | tee gpurun_out/r03D_summary.txt
import json, glob, re
def rows(f):
    out={}
    for l in open(f):
        m=re.match(r"(.{34})\s+([\d.]+) TF\s+([\d.]+) us", l)
        if m: out[m.group(1).strip()]=float(m.group(3))
    return out
a=[rows(f) for f in sorted(glob.glob("gpurun_out/r03D_gemm_prev_*.txt"))]; b=[rows(f) for f in sorted(glob.glob("gpurun_out/r03D_gemm_main_*.txt"))]
for k in a[0]:
    x=min(r[k] for r in a if k in r); y=min(r[k] for r in b if k in r)
    print(f"{k:36s} prev {x:8.1f} us   new {y:8.1f} us   {100*(x/y-1):+5.1f} %")
for f in sorted(glob.glob("gpurun_out/r03D_bench_*.json")):
    d=json.load(open(f)); r=d["roofline"]
    print(f.split("r03D_bench_")[1].ljust(18), d["value"], "img/s", d["ms_per_step"], "ms | dominant template", r["achieved"], "TF/s serial | gemm ms", r["gemm_ms_per_step"])
PY
