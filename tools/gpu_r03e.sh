#!/bin/bash
# round 3, multi-GPU readiness on one GPU: the CU-reservation probe, the 1-GPU cost of reserving CUs, the bucket exchange
# rehearsed through the C ABI's RCCL communicator
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/reserve_cus_probe.py 2>&1 | tee gpurun_out/r03e_reserve_cus_probe.txt
for rep in 1 2; do
for n in 0 8 16; do
  DEVIT_RESERVE_CUS=$n timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03e_bench_reserve${n}_$rep.json 2> gpurun_out/r03e_bench.err
done; done
for ex in abi torch; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --classes 250 --rehearse-exchange $ex > gpurun_out/r03e_bench_exchange_$ex.json 2> gpurun_out/r03e_bench_exchange_$ex.err
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --classes 250 > gpurun_out/r03e_bench_exchange_none.json 2> gpurun_out/r03e_bench.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03e_bench_*.json")):
    d=json.load(open(f))
    print(f.split("r03e_bench_")[1], d["value"], d["ms_per_step"], "reserved", d["reserved_cus"], d["rehearse_exchange"], d["allreduce_ms"], d["overlap_frac"], d["bucket_mb"])
PY
