#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="${GEMM_ARGS:-50688 2304 768 0 0 0 3}"
cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcL$i -- python3 $R/tools/gemm_one.py $ARGS > $R/gpurun_out/pmcL$i.log 2>&1
done
cd $R
for d in pmcL1 pmcL2 pmcL3 pmcL4; do f=$(find gpurun_out/$d -name "*counter_collection.csv" | head -1); [ -z "$f" ] && { echo "no csv for $d"; tail -3 gpurun_out/$d.log; continue; }; python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "gemm" in k:
        for c, v in d.items(): print("   %-28s per dispatch %16.0f" % (c, v / cnt[(k, c)]))
PY
done
