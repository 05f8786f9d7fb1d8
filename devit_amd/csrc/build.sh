#!/bin/bash
# Build libdevit_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir]
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/..}"
TOOLS="$HERE/../../tools"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -fvisibility=hidden: the library exports the C ABI of include/devit_hip.h (DEVIT_API) and nothing else
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-result"
OBJS="api gemm layernorm attention elementwise losses sgemm comm encoder shrink"
mkdir -p "$HERE/build"
# The asm K loops of the four-wave and full-row GEMM kernels are GENERATED (tools/gen_gemm4.py, tools/gen_gemmfr.py document the register
# plans): made here, not committed (2 MB of text); tests/test_abi.py regenerates them and compares with what the library was built from.
for g in gemm4 gemmfr; do
  if [ ! -f "$HERE/${g}_kloop.inc" ] || [ "$TOOLS/gen_${g}.py" -nt "$HERE/${g}_kloop.inc" ]; then
    env -u GEMMFR_NOREQ -u GEMMFR_FIXSRC -u GEMMFR_A_NT -u GEMMFR_STAMP_VM python3 "$TOOLS/gen_${g}.py" "$HERE/${g}_kloop.inc" > /dev/null
  fi
done
pids=()
for f in $OBJS; do
  if [ ! -f "$HERE/build/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/build/$f.o" ] || [ "$HERE/build.sh" -nt "$HERE/build/$f.o" ] || \
     [ "$HERE/devit_common.h" -nt "$HERE/build/$f.o" ] || { [ "$f" = gemm ] && { [ "$HERE/gemm4_kloop.inc" -nt "$HERE/build/$f.o" ] || [ "$HERE/gemmfr_kloop.inc" -nt "$HERE/build/$f.o" ]; }; } || [ "$HERE/../../include/devit_hip.h" -nt "$HERE/build/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/build/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
# Build gates over the gfx950 code objects (check_objects.py: no spills / scratch anywhere; nothing but MFMAs writes AGPRs in the kernels whose
# asm K loops leave their accumulators there).  A missing LLVM tool fails the build: a gate that cannot run is not a pass.
python3 "$HERE/check_objects.py" "$HERE/build" $OBJS
( cd "$HERE/build" && $HIPCC --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o "$OUT/libdevit_hip.so" $(for f in $OBJS; do echo $f.o; done) -ldl )
echo "built $OUT/libdevit_hip.so"
