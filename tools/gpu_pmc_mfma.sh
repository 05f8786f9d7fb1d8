#!/bin/bash
# MFMA-pipe utilisation of every kernel of the bench step: one rocprofv3 counter pass (SQ + GRBM slots only; kernel trace
# only, python3 directly after `--`), summarised per kernel template by tools/pmc_mfma.py.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/pmc_mfma
timeout 1200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --teacher-lookahead 0 > $R/gpurun_out/pmc_mfma.log 2>&1
tail -n 2 $R/gpurun_out/pmc_mfma.log | cut -c1-200
cd $R
python3 tools/pmc_mfma.py gpurun_out/pmc_mfma gpurun_out/mfma.json
find gpurun_out/pmc_mfma -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_mfma -name "*counter_collection.csv" -size +20M -delete
