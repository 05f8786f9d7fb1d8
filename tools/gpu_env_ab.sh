#!/bin/bash
# bench A/B of an environment switch in one box: tools/gpu_env_ab.sh VAR A B
export TMPDIR=/tmp
for i in 1 2 3; do for v in $2 $3; do echo "== $1=$v"; env $1=$v timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 | cut -c60-200; done; done
