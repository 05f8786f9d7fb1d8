#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for gn in 2 4 9; do
  export DEVIT_GEMM_GN=$gn
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcG$gn -- python3 $R/tools/gemm_one.py 50688 2304 768 0 0 0 3 > $R/gpurun_out/pmcG$gn.log 2>&1
  f=$(find $R/gpurun_out/pmcG$gn -name "*counter_collection.csv" | head -1)
  echo "== GN=$gn"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    if "gemm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c in agg: agg[c] /= n[c]
print({k: int(v) for k, v in agg.items()}, "hit rate %.3f" % (agg["TCC_HIT_sum"] / (agg["TCC_HIT_sum"] + agg["TCC_MISS_sum"])), "read-miss MB %.0f" % (agg["TCC_EA0_RDREQ_sum"] * 128 / 1e6))
PY
  t=$(find $R/gpurun_out/pmcG$gn -name "*kernel_trace.csv" | head -1)
  python3 - "$t" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm" in r["Kernel_Name"]]
print("durations us:", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows])
PY
done
