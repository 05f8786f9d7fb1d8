python -m pytest tests -m gpu -q 2>&1 | grep -v "^  " | tail -30
python tools/host_profile.py 2>&1 | tail -12
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err; head -c 500 gpurun_out/bench_d.json; tail -3 gpurun_out/bench_d.err
