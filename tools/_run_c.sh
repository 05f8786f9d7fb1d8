for s in "50688 2304 768 0" "50688 3072 768 1" "50688 768 3072 2" "50688 768 768 2" "50688 1536 384 1" "50688 384 1536 2" "50688 384 1536 0 1"; do
  python tools/gemm_shape.py $s
  for v in ant bnt; do DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so python tools/gemm_shape.py $s; done
done
for gn in 1 2 3 9; do echo "gn=$gn"; DEVIT_GEMM_GN=$gn python tools/gemm_shape.py 50688 2304 768 0; DEVIT_GEMM_GN=$gn python tools/gemm_shape.py 50688 3072 768 1; done
