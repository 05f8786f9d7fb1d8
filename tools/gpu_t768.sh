export TMPDIR=/tmp
python tools/gemm_t768.py 2>&1 | grep TF
DEVIT_GEMM_TILE=2 python tools/gemm_t768.py 2>&1 | grep TF
python tools/gemm_t768.py 2>&1 | grep TF
