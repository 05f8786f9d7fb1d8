#!/bin/bash
# build a diagnostic variant of the library: tools/build_variant.sh NAME "-DFLAG ..."  -> tools/_diag/libdevit_NAME.so
set -e
NAME=$1; EXTRA=$2
cd "$(dirname "$0")/../devit_amd/csrc"
mkdir -p build_$NAME ../../tools/_diag
for f in api gemm layernorm attention elementwise losses sgemm comm encoder shrink; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-result $EXTRA -c $f.hip -o build_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o ../../tools/_diag/libdevit_$NAME.so build_$NAME/*.o -ldl
echo built variant $NAME
