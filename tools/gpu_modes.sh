#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
echo "== lookahead"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 | cut -c60-170
echo "== side-stream-in-step"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --teacher-lookahead 0 2>&1 | tail -n 1 | cut -c60-170
echo "== serial"; DEVIT_TEACHER_STREAM=0 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --teacher-lookahead 0 2>&1 | tail -n 1 | cut -c60-170
done
