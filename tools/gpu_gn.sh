#!/bin/bash
export TMPDIR=/tmp
for gn in 100 8 4 2 0; do echo "== GN=$gn"; DEVIT_GEMM_GN=$gn timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "NT|dgrad"; done
