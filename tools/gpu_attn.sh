#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "attention or relation" > gpurun_out/kernels.log 2>&1; tail -n 15 gpurun_out/kernels.log
timeout 600 python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from devit_amd import ops
from devit_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda")
for H, tag in ((6, "student"), (12, "teacher")):
    B, N = 256, 198; D = H * 64; M = B * N
    qkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev, extra=128); qkv[:M] = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
    out = ops.rows_alloc(M, D, torch.bfloat16, dev); lse = torch.empty(B, H, N, device=dev)
    dout = ops.rows_alloc(M, D, torch.bfloat16, dev); dout[:M] = torch.randn(M, D, device=dev).to(torch.bfloat16)
    dqkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev)
    def t(fn, reps=10):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [fn() for _ in range(reps)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
    f = t(lambda: call("devit_attn_fwd", ptr(qkv), ptr(out), ptr(lse), None, B, N, H, 64, 0.125, stream_ptr()))
    b = t(lambda: call("devit_attn_bwd", ptr(qkv), ptr(out), ptr(dout), ptr(lse), None, None, ptr(dqkv), B, N, H, 64, 0.125, stream_ptr()))
    print(f"{tag}: attn fwd {f:.1f} us ({(M*3*D+M*D)*2/f/1e6:.2f} TB/s)   bwd {b:.1f} us ({(M*3*D*2+M*D*2)*2/b/1e6:.2f} TB/s)")
PY
