#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
for f in 0 1 3; do echo "== force$f"; DEVIT_GEMM_FORCE=$f timeout 300 python tools/gemm_tiles.py 2>&1 | grep TF; done
done
