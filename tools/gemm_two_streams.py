#!/usr/bin/env python3
"""Two GEMM launches: back to back on one stream vs at the same time on two streams (how much of one kernel's partial
last round does the other fill?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda"); BF = torch.bfloat16; M = 50688
def mk(N, K, kind):
    a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * .02).to(BF)
    out = torch.empty(M, N, dtype=torch.float32 if kind == 2 else BF, device=dev)
    res = torch.randn(M, N, device=dev) if kind == 2 else None
    return lambda: ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, res=res)
pairs = [("T fc2 + S qkv", mk(768, 3072, 2), mk(1152, 384, 0)), ("T proj + S fc2", mk(768, 768, 2), mk(384, 1536, 2)),
         ("T qkv + S fc1", mk(2304, 768, 0), mk(1536, 384, 1))]
s2 = torch.cuda.Stream()
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for name, fa, fb in pairs:
    ta, tb = timed(fa), timed(fb)
    def both():
        s2.wait_stream(torch.cuda.current_stream())
        fa()
        with torch.cuda.stream(s2): fb()
        torch.cuda.current_stream().wait_stream(s2)
    tc = timed(both)
    print(f"{name:16s} alone {ta:6.1f} + {tb:6.1f} = {ta+tb:6.1f} us | concurrent {tc:6.1f} us")
