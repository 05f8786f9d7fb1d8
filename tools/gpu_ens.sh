#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -k "ensemble" > gpurun_out/ens.log 2>&1; tail -n 30 gpurun_out/ens.log
