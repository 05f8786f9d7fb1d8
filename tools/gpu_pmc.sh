#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/tools/gemm_one.py ${SHAPE:-50688 2304 768 0 0 0 3} > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/tools/gemm_one.py ${SHAPE:-50688 2304 768 0 0 0 3} > $R/gpurun_out/pmc2.log 2>&1
cd $R
for d in pmc1 pmc2; do f=$(find gpurun_out/$d -name "*counter_collection.csv" | head -1); echo $f; python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "gemm" in k:
        print(k)
        for c, v in d.items(): print("   %-28s %16.0f  (per dispatch %14.0f)" % (c, v, v / cnt[(k, c)]))
PY
done
tail -3 gpurun_out/pmc1.log
