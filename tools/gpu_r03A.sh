#!/bin/bash
# experiment: fp32 epilogues with whole 128-byte rows per instruction (gemm.hip DEVIT_F32_FULL_ROWS) -- correctness, then A/B against the 64-byte form
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_lean.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r03A_tests.txt 2>&1 || { tail -30 gpurun_out/r03A_tests.txt; exit 1; }
tail -3 gpurun_out/r03A_tests.txt
timeout -k 10 300 python tools/gemm_race_screen.py 12 > gpurun_out/r03A_race.txt 2>&1 || { tail -20 gpurun_out/r03A_race.txt; exit 1; }
tail -2 gpurun_out/r03A_race.txt
for rep in 1 2; do for v in rows0 main; do
  if [ $v = main ]; then unset DEVIT_LIB_PATH; else export DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so; fi
  COLD=1 timeout -k 10 300 python tools/gemm_bench.py > gpurun_out/r03A_gemm_${v}_$rep.txt 2>&1
done; done
unset DEVIT_LIB_PATH
for rep in 1 2 3; do for v in rows0 main; do
  if [ $v = main ]; then unset DEVIT_LIB_PATH; else export DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03A_bench_${v}_$rep.json 2> gpurun_out/r03A_bench.err
done; done
unset DEVIT_LIB_PATH
python - <<'PY' | tee gpurun_out/r03A_summary.txt
import json, glob, re
def rows(f):
    out={}
    for l in open(f):
        m=re.match(r"(.{34})\s+([\d.]+) TF\s+([\d.]+) us", l)
        if m: out[m.group(1).strip()]=float(m.group(3))
    return out
a=[rows(f) for f in sorted(glob.glob("gpurun_out/r03A_gemm_rows0_*.txt"))]; b=[rows(f) for f in sorted(glob.glob("gpurun_out/r03A_gemm_main_*.txt"))]
for k in a[0]:
    x=min(r[k] for r in a if k in r); y=min(r[k] for r in b if k in r)
    print(f"{k:36s} off {x:8.1f} us   128-B {y:8.1f} us   {100*(x/y-1):+5.1f} %")
for f in sorted(glob.glob("gpurun_out/r03A_bench_*.json")):
    d=json.load(open(f)); r=d["roofline"]
    print(f.split("r03A_bench_")[1].ljust(18), d["value"], "img/s", d["ms_per_step"], "ms | dominant template", r["achieved"], "TF/s serial | gemm ms", r["gemm_ms_per_step"])
PY
