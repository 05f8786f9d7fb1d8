#!/bin/bash
# Build libdevit_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir]
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
mkdir -p "$HERE/build"
pids=()
for f in api gemm layernorm attention elementwise losses sgemm comm encoder shrink; do
  if [ ! -f "$HERE/build/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/build/$f.o" ] || \
     [ "$HERE/devit_common.h" -nt "$HERE/build/$f.o" ] || { [ "$f" = gemm ] && [ "$HERE/gemm4_kloop.inc" -nt "$HERE/build/$f.o" ]; } || [ "$HERE/../../include/devit_hip.h" -nt "$HERE/build/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/build/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libdevit_hip.so" "$HERE"/build/{api,gemm,layernorm,attention,elementwise,losses,sgemm,comm,encoder,shrink}.o -ldl
echo "built $OUT/libdevit_hip.so"
