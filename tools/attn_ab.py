#!/usr/bin/env python3
"""Cold-cache timing of attention forward / backward at B = 256 (student H = 6 and teacher H = 12 forward; student backward):
median of 9 launches, Infinity Cache flushed before each.  One library per process (DEVIT_LIB_PATH)."""
import os, sys, statistics as st, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops
from devit_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda"); B, N = 256, 198
flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
tag = os.path.basename(os.environ.get("DEVIT_LIB_PATH", "default")).replace("libdevit_", "").replace(".so", "")
def cold(fn, reps=9):
    fn(); ts = []
    for _ in range(reps):
        flush.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return st.median(ts)
res = []
for H in (6, 12):
    D = H * 64; M = B * N
    qkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev, extra=128); qkv[:M] = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
    out = ops.rows_alloc(M, D, torch.bfloat16, dev); lse = torch.empty(B, H, N, device=dev)
    dout = ops.rows_alloc(M, D, torch.bfloat16, dev); dout[:M] = torch.randn(M, D, device=dev).to(torch.bfloat16)
    dqkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev)
    fwd = lambda: call("devit_attn_fwd", ptr(qkv), ptr(out), ptr(lse), None, B, N, H, 64, 0.125, 0, stream_ptr())
    res.append(f"fwd H={H:2d} {cold(fwd):6.1f}")
    if H == 6:
        bwd = lambda: call("devit_attn_bwd", ptr(qkv), ptr(out), ptr(dout), ptr(lse), None, None, ptr(dqkv), B, N, H, 64, 0.125, stream_ptr())
        res.append(f"bwd H={H:2d} {cold(bwd):6.1f}")
print(f"{tag:8s} " + " | ".join(res) + "   (us, median of 9, cold)", flush=True)
