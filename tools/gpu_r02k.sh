#!/bin/bash
# round-2 call k: final verification of the tree -- every gpu test, smoke(), python bench.py (defaults, with both CPU legs)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests.log 2>&1; tail -n 3 gpurun_out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 900 gpurun_out/bench_final.json
