#!/bin/bash
# full GPU suite + the default bench line
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r03b_gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err
cut -c1-1500 gpurun_out/r03b_bench.json
