#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -n 3
F='split_k=( 7|28| 56|113|  9| 37)'
for i in 1 2; do for pf in 1 0; do echo "== pf$pf"; COLD=1 DEVIT_GEMM_PF=$pf timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "TF" | grep -vE "$F"; done; done
