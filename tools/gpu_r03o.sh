#!/bin/bash
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_cli.py tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r03o_tests.log
