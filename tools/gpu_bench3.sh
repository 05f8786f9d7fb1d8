#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3; do timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 | cut -c1-200; done
