#!/bin/bash
set -eo pipefail
mkdir -p gpurun_out
DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_attnstamp.so timeout -k 5 120 python tools/attn_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03k_attn_stamps.txt
for rep in 1 2; do
  DEVIT_ATTN_BWD=4 timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/bwd4 /' | tee -a gpurun_out/r03k_attn_ab.txt
done
