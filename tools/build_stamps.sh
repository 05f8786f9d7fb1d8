#!/bin/bash
# Diagnostic-only build: libdevit_hip_stamps.so with s_memtime stamps in the GEMM K-loop.
set -e
cd "$(dirname "$0")/../devit_amd/csrc"
mkdir -p build_stamps
for f in api gemm layernorm attention elementwise losses sgemm; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DDEVIT_GEMM_STAMPS $EXTRA -c $f.hip -o build_stamps/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_diag/libdevit_hip_stamps.so build_stamps/*.o
echo built stamps lib
