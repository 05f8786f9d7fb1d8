#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/kernels.log 2>&1; tail -n 3 gpurun_out/kernels.log
timeout 600 python tools/gemm_bench.py 2>&1 | grep TF
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -n 1 gpurun_out/bench.log | cut -c1-400
