#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
DEVIT_TEACHER_STREAM=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stream=0', d['value'], d['ms_per_step'])"
DEVIT_TEACHER_STREAM=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stream=1', d['value'], d['ms_per_step'])"
done
