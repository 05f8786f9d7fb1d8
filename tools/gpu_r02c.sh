#!/bin/bash
# round-2 call c: correctness of the GEMM with the de-phasing switch on (full-size bit-exact tests), then the A/B sweep
mkdir -p gpurun_out; export TMPDIR=/tmp
DEVIT_GEMM_STAGGER=40000,2 timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm" > gpurun_out/tests_stagger.log 2>&1; tail -n 3 gpurun_out/tests_stagger.log
timeout 900 python tools/gemm_stagger.py > gpurun_out/stagger.txt 2>&1; tail -n 40 gpurun_out/stagger.txt
