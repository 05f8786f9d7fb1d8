#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() { # name args...
  name=$1; shift
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/tools/gemm_one.py "$@" > $R/gpurun_out/pmc_$name.log 2>&1
  f=$(find $R/gpurun_out/pmc_$name -name "*counter_collection.csv" | head -1)
  echo "== $name $@"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    if "gemm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print({k: int(v / n[k]) for k, v in agg.items()})
PY
}
run nt 50688 2304 768 0 0 0 2
run dgrad 50688 384 1536 0 0 1 2
run wgrad 1536 384 50688 5 1 1 2
run wgrad128 1152 384 50688 5 1 1 2
