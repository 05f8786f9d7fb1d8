#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -n 4
bash tools/gpu_ab6.sh "$@" | grep -E "==|T |S fc1  NT|dgelu"
