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
     [ "$HERE/devit_common.h" -nt "$HERE/build/$f.o" ] || { [ "$f" = gemm ] && { [ "$HERE/gemm4_kloop.inc" -nt "$HERE/build/$f.o" ] || [ "$HERE/gemmfr_kloop.inc" -nt "$HERE/build/$f.o" ]; }; } || [ "$HERE/../../include/devit_hip.h" -nt "$HERE/build/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/build/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
# No kernel of the library may spill registers: a scratch reload inside a GEMM K-step is a vector-memory operation that lands in
# the kernel's own counted vmcnt waits.  The per-kernel metadata of every object is checked here.
for f in api gemm layernorm attention elementwise losses sgemm comm encoder shrink; do
  LLVM=/opt/rocm/lib/llvm/bin
  $LLVM/llvm-objcopy --dump-section .hip_fatbin="$HERE/build/$f.fatbin" "$HERE/build/$f.o" 2>/dev/null || continue   # (no device code)
  $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$HERE/build/$f.fatbin" \
      --output="$HERE/build/$f.gfx950.co" --unbundle || { echo "build.sh: cannot unbundle $f.o"; exit 1; }
  rm -f "$HERE/build/$f.fatbin"
  bad=$($LLVM/llvm-readelf --notes "$HERE/build/$f.gfx950.co" 2>/dev/null | \
        awk '/\.name:/{n=$2} /\.vgpr_spill_count:/{if ($2+0 > 0) print n" spills "$2" VGPRs"} /\.private_segment_fixed_size:/{if ($2+0 > 0) print n" uses "$2" B of scratch"}')
  rm -f "$HERE/build/$f.gfx950.co"
  if [ -n "$bad" ]; then echo "build.sh: register spills in $f.hip:"; echo "$bad"; exit 1; fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libdevit_hip.so" "$HERE"/build/{api,gemm,layernorm,attention,elementwise,losses,sgemm,comm,encoder,shrink}.o -ldl
echo "built $OUT/libdevit_hip.so"
