#!/bin/bash
# round-3 call q: final tree -- every gpu test, smoke(), the round's profile set, the secondary configs
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests.log 2>&1; tail -n 3 gpurun_out/tests.log
grep -q " passed" gpurun_out/tests.log && ! grep -q " failed" gpurun_out/tests.log || exit 1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
bash tools/gpu_profile_round.sh
bash tools/gpu_pmc_mfma.sh
bash tools/gpu_pmc_traffic.sh
timeout 600 python tools/bench_configs.py 2>/dev/null | tee gpurun_out/bench_configs.jsonl
