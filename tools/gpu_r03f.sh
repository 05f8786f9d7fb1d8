#!/bin/bash
# multi-GPU readiness on one GPU, final form: probe (warmed up, each setting twice), 1-GPU cost of reserving CUs at all times,
# and the exchange rehearsed through the C ABI's RCCL communicator with the scoped reservation (16 CUs while buckets fly)
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/reserve_cus_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03f_reserve_cus_probe.txt
for rep in 1 2; do
  for n in 0 16 32; do
    DEVIT_RESERVE_CUS=$n timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --classes 250 > gpurun_out/r03f_bench_always${n}_$rep.json 2> gpurun_out/r03f_bench.err
  done
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --classes 250 --rehearse-exchange abi > gpurun_out/r03f_bench_abi_scoped16_$rep.json 2> gpurun_out/r03f_bench.err
  DEVIT_RESERVE_CUS=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --classes 250 --rehearse-exchange abi > gpurun_out/r03f_bench_abi_noreserve_$rep.json 2> gpurun_out/r03f_bench.err
done
python - <<'PY' | tee gpurun_out/r03f_summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r03f_bench_*.json")):
    d=json.load(open(f))
    print(f.split("r03f_bench_")[1].ljust(28), d["value"], "img/s", d["ms_per_step"], "ms | reserved always", d["reserved_cus"], "| while buckets fly", d["reserved_cus_while_buckets_in_flight"], "| exchange", d["rehearse_exchange"], d["allreduce_ms"], d["overlap_frac"])
PY
