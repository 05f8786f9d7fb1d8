#!/bin/bash
# four-wave GEMM A/B on ONE box: tools/gemm_bench.py COLD=1 with DEVIT_GEMM4=0 / 1 interleaved (teacher + student fc1 lines), then bench.py
export TMPDIR=/tmp; mkdir -p gpurun_out
F='T qkv|T fc1|T fc2|T proj|S fc1  NT|S qkv  NT'
for i in 1 2; do
  for v in 0 1; do echo "== DEVIT_GEMM4=$v"; DEVIT_GEMM4=$v COLD=1 timeout -k 10 300 python tools/gemm_bench.py 2>&1 | grep -E "$F" || exit 1; done
done
for i in 1 2; do
  for v in 0 1; do echo "== bench DEVIT_GEMM4=$v"; DEVIT_GEMM4=$v timeout -k 10 600 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'dominant', r['achieved'], 'gemm_ms', r['gemm_ms_per_step'])" || exit 1; done
done
