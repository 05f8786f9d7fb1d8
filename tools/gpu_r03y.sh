#!/bin/bash
# experiment: four-wave 256x256 GEMM kernel (DEVIT_GEMM4=1) -- correctness, then A/B against the 8-wave ping-pong kernel
set -eo pipefail
mkdir -p gpurun_out
DEVIT_GEMM4=1 timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "gemm or fullsize or step" > gpurun_out/r03y_tests.txt 2>&1 || { tail -30 gpurun_out/r03y_tests.txt; exit 1; }
tail -3 gpurun_out/r03y_tests.txt
for rep in 1 2; do for v in 0 1; do
  DEVIT_GEMM4=$v COLD=1 timeout -k 10 300 python tools/gemm_bench.py > gpurun_out/r03y_gemm_${v}_$rep.txt 2>&1
done; done
for rep in 1 2; do for v in 0 1; do
  DEVIT_GEMM4=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03y_bench_${v}_$rep.json 2> gpurun_out/r03y_bench.err
done; done
python - <<'PY' | tee gpurun_out/r03y_summary.txt
import json, glob, re
def rows(f):
    out={}
    for l in open(f):
        m=re.match(r"(.{34})\s+([\d.]+) TF\s+([\d.]+) us", l)
        if m: out[m.group(1).strip()]=float(m.group(3))
    return out
a=[rows(f) for f in sorted(glob.glob("gpurun_out/r03y_gemm_0_*.txt"))]; b=[rows(f) for f in sorted(glob.glob("gpurun_out/r03y_gemm_1_*.txt"))]
for k in a[0]:
    x=min(r[k] for r in a if k in r); y=min(r[k] for r in b if k in r)
    print(f"{k:36s} 8-wave {x:8.1f} us   4-wave {y:8.1f} us   {100*(x/y-1):+5.1f} %")
for f in sorted(glob.glob("gpurun_out/r03y_bench_*.json")):
    d=json.load(open(f)); r=d["roofline"]
    print(f.split("r03y_bench_")[1].ljust(12), d["value"], "img/s", d["ms_per_step"], "ms | dominant template", r["achieved"], "TF/s serial | gemm ms", r["gemm_ms_per_step"])
PY
