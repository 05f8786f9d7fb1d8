#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -n 3
F='split_k=( 7|28| 56|113|  9| 37)'
for c in 1 0; do for i in 1 2; do
echo "== new$c"; COLD=$c timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "TF" | grep -vE "$F"
echo "== prev$c"; COLD=$c DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_prev.so timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "TF" | grep -vE "$F"
done; done
