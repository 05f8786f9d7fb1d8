#!/bin/bash
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_lean.py -x -q -m gpu -k "attention or attn or lean" 2>&1 | tail -3 | tee gpurun_out/r03j_tests.log
DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_attnstamp.so timeout -k 5 120 python tools/attn_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03j_attn_stamps.txt
for rep in 1 2; do
  DEVIT_ATTN_BWD=8 timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/bwd8 /' | tee -a gpurun_out/r03j_attn_ab.txt
  DEVIT_ATTN_BWD=4 timeout -k 10 120 python tools/attn_ab.py 2>&1 | tail -1 | sed 's/^/bwd4 /' | tee -a gpurun_out/r03j_attn_ab.txt
done
