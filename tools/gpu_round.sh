#!/bin/bash
# One parameterised call script for a GPU box (replaces the per-experiment gpu_rNN*.sh files of rounds 2-3).
#   bash tools/gpu_round.sh [tests] [smoke] [bench] [profile] [pmc] [configs] [ab:V1,V2] [gemm:V1,V2] [py:script.py]
# Steps run in the order given and the script stops at the first failing step (never start a GPU step after one that was killed).
#   tests    every `-m gpu` test (TESTS="tests/test_gpu_kernels.py -k gemm" narrows it)
#   smoke    __graft_entry__.smoke()
#   bench    one short bench.py line (no CPU leg)
#   profile  tools/gpu_profile_round.sh: bench.py with the CPU leg + rocprofv3 --kernel-trace --stats (two-stream and serialized)
#   pmc      tools/gpu_pmc_mfma.sh + tools/gpu_pmc_traffic.sh (separate counter passes)
#   configs  tools/bench_configs.py (BASELINE secondary configs)
#   ab:V,..  bench.py interleaved: default library vs tools/_diag/libdevit_V.so (tools/build_variant.sh) on THIS box
#   gemm:V,..tools/gemm_bench.py COLD=1 interleaved the same way
#   py:F     timeout 600 python F
mkdir -p gpurun_out; export TMPDIR=/tmp
for step in "$@"; do
  echo "=== $step"
  case "$step" in
    tests)
      timeout 1100 python -m pytest ${TESTS:-tests} -m gpu -q --tb=short -x -p no:cacheprovider > gpurun_out/tests.log 2>&1
      rc=$?; tail -n 15 gpurun_out/tests.log; [ $rc -eq 0 ] || exit 1 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2 || exit 1 ;;
    bench)
      timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1 || { tail -n 20 gpurun_out/bench.log; exit 1; }
      tail -n 1 gpurun_out/bench.log > gpurun_out/bench_short.json; cut -c1-400 gpurun_out/bench_short.json ;;
    profile) bash tools/gpu_profile_round.sh || exit 1 ;;
    pmc) bash tools/gpu_pmc_mfma.sh && bash tools/gpu_pmc_traffic.sh || exit 1 ;;
    configs) timeout 600 python tools/bench_configs.py 2>/dev/null | tee gpurun_out/bench_configs.jsonl || exit 1 ;;
    ab:*) bash tools/gpu_bench_ab.sh $(echo "${step#ab:}" | tr ',' ' ') 2>&1 | tee gpurun_out/bench_ab.txt || exit 1 ;;
    gemm:*) COLD=1 bash tools/gpu_ab6.sh $(echo "${step#gemm:}" | tr ',' ' ') 2>&1 | tee gpurun_out/gemm_ab.txt || exit 1 ;;
    py:*) timeout 600 python "${step#py:}" 2>&1 | tee -a gpurun_out/py.txt || exit 1 ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
