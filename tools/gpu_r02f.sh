#!/bin/bash
# round-2 call f: world-size-2 rehearsal of the real step on one GPU (gloo transport), and config 4's per-GPU workload
# (C = 250, bs 256) at N = 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ddp.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_ddp.log 2>&1; tail -n 25 gpurun_out/tests_ddp.log
cat gpurun_out/ddp_two_ranks_one_gpu.json 2>/dev/null
timeout 600 python bench.py --classes 250 --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/bench_c250.err | tail -n 1 > gpurun_out/bench_c250.json
python3 -c "import json; d=json.load(open('gpurun_out/bench_c250.json')); print('C=250 N=1:', d['value'], 'img/s', d['ms_per_step'], 'ms', d['config']['workload'][:80])"
