#!/bin/bash
# bench A/B in one box: default library vs tools/_diag/libdevit_$v.so for each argument
export TMPDIR=/tmp
for i in 1 2 3; do
echo "== new"; timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 | cut -c60-200
for v in "$@"; do echo "== $v"; DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 | cut -c60-200; done
done
