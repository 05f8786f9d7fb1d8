#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_LDS_UNALIGNED_STALL SQ_WAVES"; do
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcA -- python3 $R/tools/attn_one.py 6 > $R/gpurun_out/pmcA.log 2>&1
  f=$(find $R/gpurun_out/pmcA -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = "fwd" if "attn_fwd" in r["Kernel_Name"] else ("bwd" if "attn_bwd" in r["Kernel_Name"] else None)
    if k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in agg: print(k, {c: int(v / n[(k, c)]) for c, v in agg[k].items()})
PY
  rm -rf $R/gpurun_out/pmcA
done
