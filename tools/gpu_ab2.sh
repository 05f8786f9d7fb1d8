#!/bin/bash
export TMPDIR=/tmp
V=${1:-dmamid}
for i in 1 2; do
echo "== base"; timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "TF" | grep -vE "split_k=( 8|16|32|57)"
echo "== $V"; DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$V.so timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "TF" | grep -vE "split_k=( 8|16|32|57)"
done
