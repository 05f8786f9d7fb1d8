#!/bin/bash
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "compacted or gate_reassign or losses_vs_golden" 2>&1 | tail -30 | tee gpurun_out/r03c_tests.log
