#!/bin/bash
# One GPU round: kernel tests, model tests, smoke.  Output -> gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernels exit $?" >> gpurun_out/kernels.log
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/model.log 2>&1
echo "model exit $?" >> gpurun_out/model.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
tail -n 40 gpurun_out/kernels.log
tail -n 30 gpurun_out/model.log
tail -n 5 gpurun_out/smoke.log
