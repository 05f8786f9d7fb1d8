#!/bin/bash
# round-2 call d: the round's artefacts for the current tree -- bench (with CPU baseline), rocprofv3 kernel stats (two-stream
# and serialized), then the PMC passes (MFMA busy, traffic) in their own runs (kernel-trace only)
bash tools/gpu_profile_round.sh
bash tools/gpu_pmc_mfma.sh
bash tools/gpu_pmc_traffic.sh
ls gpurun_out | head -40
