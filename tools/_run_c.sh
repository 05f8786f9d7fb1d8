timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -x -k "gemm or wgrad" 2>&1 | grep -v "^  " | tail -4
timeout 600 python tools/gemm_race_screen.py 10 2>&1 | tail -3
for rep in 1 2; do
for s in "50688 2304 768 0" "50688 3072 768 1" "50688 768 3072 2" "50688 1536 384 1" "50688 1536 384 4 1"; do
  python tools/gemm_shape.py $s
  for v in nosplit; do DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so python tools/gemm_shape.py $s; done
done; done
