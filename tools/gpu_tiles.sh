export TMPDIR=/tmp
for r in 1 2; do for t in 0 1 2; do if [ $t = 0 ]; then python tools/gemm_tiles.py; else DEVIT_GEMM_TILE=$t python tools/gemm_tiles.py; fi; done; done 2>&1 | grep TF
