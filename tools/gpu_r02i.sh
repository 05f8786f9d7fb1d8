#!/bin/bash
# round-2 call i: ping-pong read-interval variants (DMA pairs interleaved with the fragment reads; wave priority), same box,
# interleaved processes
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 200 python tools/gemm_pp_shapes.py 2>/dev/null
  for v in il p1 p2 ilp1 ilp2; do DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so timeout 200 python tools/gemm_pp_shapes.py 2>/dev/null; done
done | tee gpurun_out/pp_variants.txt
