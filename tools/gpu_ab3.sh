#!/bin/bash
export TMPDIR=/tmp
V=${1:-prev}
for i in 1 2; do
echo "== new"; timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "TF" | grep -vE "split_k=( 7|28| 56|113|  9| 37)"
echo "== $V"; DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$V.so timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "TF" | grep -vE "split_k=( 7|28| 56|113|  9| 37)"
done
