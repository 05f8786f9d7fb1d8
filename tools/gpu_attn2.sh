#!/bin/bash
export TMPDIR=/tmp
run() { timeout 300 python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from devit_amd import ops
from devit_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda")
for H, tag in ((6, "student"),):
    B, N = 256, 198; D = H * 64; M = B * N
    qkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev, extra=128); qkv[:M] = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
    out = ops.rows_alloc(M, D, torch.bfloat16, dev); lse = torch.empty(B, H, N, device=dev)
    dout = ops.rows_alloc(M, D, torch.bfloat16, dev); dout[:M] = torch.randn(M, D, device=dev).to(torch.bfloat16)
    dqkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev)
    big = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
    def t(fn, reps=10):
        best = 1e9
        for _ in range(reps):
            big.zero_()    # flush the Infinity Cache: in the step the operands come from HBM
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e3)
        return best
    call("devit_attn_fwd", ptr(qkv), ptr(out), ptr(lse), None, B, N, H, 64, 0.125, stream_ptr())
    f = t(lambda: call("devit_attn_fwd", ptr(qkv), ptr(out), ptr(lse), None, B, N, H, 64, 0.125, stream_ptr()))
    b = t(lambda: call("devit_attn_bwd", ptr(qkv), ptr(out), ptr(dout), ptr(lse), None, None, ptr(dqkv), B, N, H, 64, 0.125, stream_ptr()))
    print(f"{tag}: attn fwd {f:.1f} us   bwd {b:.1f} us (cold caches)")
PY
}
for i in 1 2; do echo "== new"; run; for v in "$@"; do echo "== $v"; DEVIT_LIB_PATH=$PWD/tools/_diag/libdevit_$v.so run; done; done
