#!/bin/bash
# bench.py --shrink 0.3, masked vs compact, two interleaved pairs (value ms_per_step)
export TMPDIR=/tmp
for i in 1 2; do for m in masked compact; do
  echo -n "shrink 0.3 $m "; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shrink 0.3 --shrink-mode $m 2>&1 | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
