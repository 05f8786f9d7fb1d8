export TMPDIR=/tmp
python tools/gemm_bench.py 2>&1 | grep -E "wgrad|dgrad"
echo "== wgrad tile 1 (sk chosen for 2 WG/CU would differ; same sk here)"
DEVIT_GEMM_WGRAD_TILE=1 python tools/gemm_bench.py 2>&1 | grep -E "wgrad"
