#!/bin/bash
# round-2 call j: verification of the current tree -- every gpu test, smoke(), the single-rank RCCL path, secondary configs
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests.log 2>&1; tail -n 8 gpurun_out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
timeout 300 python tools/rccl_smoke.py > gpurun_out/rccl_smoke.txt 2>&1; tail -n 6 gpurun_out/rccl_smoke.txt
timeout 600 python tools/bench_configs.py 2>/dev/null | tee gpurun_out/bench_configs.jsonl
