#!/bin/bash
# round-2 call b: GPU suite after the call-site fix, f16 error localisation, in-kernel timelines of the dominant GEMM shapes
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests.log 2>&1; tail -n 6 gpurun_out/tests.log
timeout 600 python tools/f16_localise.py > gpurun_out/f16_localise.txt 2>&1; tail -n 32 gpurun_out/f16_localise.txt
for shape in "2304 768 0" "3072 768 1" "768 3072 2"; do
  echo "=== tstamp N K kind = $shape"
  DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_tstamp.so timeout 300 python tools/gemm_tstamps.py $shape 2>&1 | tail -n 30
done > gpurun_out/tstamps.txt 2>&1
echo "=== stamp 2304 768" > gpurun_out/stamps.txt
DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_stamp.so timeout 300 python tools/gemm_stamps.py 2304 768 >> gpurun_out/stamps.txt 2>&1
tail -n 5 gpurun_out/tstamps.txt
