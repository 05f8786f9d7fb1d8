#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
for la in 1 0; do echo "lookahead=$la"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --teacher-lookahead $la 2>&1 | tail -n 1 | cut -c60-200; done
done
