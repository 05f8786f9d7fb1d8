set -eo pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do for v in main qrows0 orows0; do
  if [ $v = main ]; then unset DEVIT_LIB_PATH; else export DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03B3_bench_${v}_$rep.json 2> gpurun_out/r03B3.err
done; done
python - <<'PY' | tee gpurun_out/r03B_three_way_in_step.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r03B3_bench_*.json")):
    d=json.load(open(f)); h=d["roofline"]["hbm_bound_kernels"]
    print(f.split("r03B3_bench_")[1].ljust(18), d["value"], "img/s", d["ms_per_step"], "ms | attention fwd", h["attention_fwd"]["ms_per_step"], "bwd", h["attention_bwd"]["ms_per_step"])
PY
