#!/usr/bin/env python3
"""Attention forward: the one-item-per-workgroup kernel (DEVIT_ATTN_FWD_PERSIST=0) against the persistent double-buffered one (=1) at B = 256,
student (H = 6) and teacher (H = 12): outputs and log-sum-exp bit-identical, microseconds cold (Infinity Cache flushed) and warm, interleaved."""
import os, sys, statistics as st, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops
from devit_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda"); B, N = 256, 198
flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
def cold(fn, reps=7):
    fn(); ts = []
    for _ in range(reps):
        flush.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return st.median(ts)
def warm(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for H in (6, 12):
    D = H * 64; M = B * N
    qkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev, extra=128); qkv[:M] = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
    gate = (torch.rand(H, device=dev) > 0.2).float()
    outs = {}
    for flag in ("0", "1"):
        os.environ["DEVIT_ATTN_FWD_PERSIST"] = flag
        out = ops.rows_alloc(M, D, torch.bfloat16, dev); lse = torch.zeros(B, H, N, device=dev)
        call("devit_attn_fwd", ptr(qkv), ptr(out), ptr(lse), ptr(gate), B, N, H, 64, 0.125, 0, stream_ptr()); torch.cuda.synchronize()
        outs[flag] = (out.clone(), lse.clone())
    same = torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1])
    r = {}
    for rep in range(2):
        for flag in ("0", "1"):
            os.environ["DEVIT_ATTN_FWD_PERSIST"] = flag
            fn = lambda: call("devit_attn_fwd", ptr(qkv), ptr(out), ptr(lse), None, B, N, H, 64, 0.125, 0, stream_ptr())
            c, w = cold(fn), warm(fn)
            r[flag] = (min(r.get(flag, (1e9, 1e9))[0], c), min(r.get(flag, (1e9, 1e9))[1], w))
    print(f"H={H:2d}  bit-identical {same}   one item per workgroup: cold {r['0'][0]:6.1f} warm {r['0'][1]:6.1f}   persistent: cold {r['1'][0]:6.1f} warm {r['1'][1]:6.1f}  (us)", flush=True)
