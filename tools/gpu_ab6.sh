#!/bin/bash
export TMPDIR=/tmp
F='split_k=( 7|28| 56|113|  9| 37)'
for i in 1 2; do
echo "== new"; timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "TF" | grep -vE "$F"
for v in "$@"; do echo "== $v"; DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "TF" | grep -vE "$F"; done
done
