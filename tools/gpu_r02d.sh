#!/bin/bash
# round-2 call d: the round's artefacts for the current tree -- bench (with CPU baseline), rocprofv3 kernel stats (two-stream
# and serialized), then the PMC passes (MFMA busy, traffic) in their own runs (kernel-trace only)
bash tools/gpu_profile_round.sh
bash tools/gpu_pmc_mfma.sh
bash tools/gpu_pmc_traffic.sh
ls gpurun_out | head -40
# same-box A/B of the teacher's 16-bit type (interleaved; the headline config is bf16, BASELINE.json)
for i in 1 2; do for tp in bf16 f16; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --teacher-precision $tp 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('teacher', d['teacher_dtype'], d['value'], 'img/s', d['ms_per_step'], 'ms')"
done; done | tee gpurun_out/teacher_dtype_ab.txt
