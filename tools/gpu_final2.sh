#!/bin/bash
# round-end evidence, second call: counter passes, secondary configs, shrunk-student bench, in-step GEMM table, the exchange rehearsed at N = 1
export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/gpu_pmc_mfma.sh && bash tools/gpu_pmc_traffic.sh || exit 1
timeout 600 python tools/bench_configs.py 2>/dev/null > gpurun_out/bench_configs.jsonl; cut -c1-200 gpurun_out/bench_configs.jsonl
bash tools/gpu_shrink.sh > gpurun_out/bench_shrink.txt 2>&1; cat gpurun_out/bench_shrink.txt
timeout 300 python tools/step_gemm_table.py > gpurun_out/step_gemm_table.txt 2>&1; tail -3 gpurun_out/step_gemm_table.txt
for m in none torch abi; do echo -n "rehearse-exchange $m: "; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --rehearse-exchange $m 2>&1 | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'wgrad_groups', d['wgrad_groups'], 'buckets', d['bucket_mb'], 'allreduce_ms', d['allreduce_ms'], 'overlap', d['overlap_frac'])"; done > gpurun_out/rehearse.txt 2>&1; cat gpurun_out/rehearse.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --classes 250 2>&1 | tail -n 1 > gpurun_out/bench_c250_n1.json; cut -c1-160 gpurun_out/bench_c250_n1.json
