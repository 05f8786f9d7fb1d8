#!/bin/bash
# bench.py A/B over environment settings on ONE box, three interleaved rounds:  bash tools/gpu_env_ab.sh "VAR=a" "VAR=b OTHER=c" ...
# ("-" = the default environment).  Prints value / ms per step and the serialized families of every run.
export TMPDIR=/tmp
for i in 1 2 3; do
  for setting in "$@"; do
    [ "$setting" = "-" ] && setting=""
    echo -n "== [$setting] "
    env $setting timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>&1 | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; f=r['families']
print(d['value'], d['ms_per_step'], 'serial', f['serialized_step_ms'], 'student', f['student_ms_per_step'], {k:(v['ms_per_step'],v['tflops']) for k,v in r['other_templates'].items()})"
  done
done
