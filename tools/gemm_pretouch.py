#!/usr/bin/env python3
"""Does a sequential pre-read of the streamed operand (pulling it into the Infinity Cache) pay for itself?
cold GEMM  vs  (sequential read of A) + GEMM, both after a cache flush."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda"); BF = torch.bfloat16; M = 50688
big = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
def t(fn):
    best = 1e9
    for _ in range(6):
        big.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
for name, N, K, bkm in (("S qkv", 1152, 384, 0), ("S fc2(store)", 384, 1536, 0), ("S fc1 dgrad", 384, 1536, 1), ("S qkv dgrad", 384, 1152, 1), ("T qkv", 2304, 768, 0)):
    a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn((K, N) if bkm else (N, K), device=dev) * .02).to(BF)
    out = torch.empty(M, N, dtype=BF, device=dev)
    g = lambda: ops.gemm(a, K, 0, w, N if bkm else K, bkm, M, N, K, kind=0, out=out, ldc=N)
    a32 = a.view(torch.int32)
    touch = lambda: torch.sum(a32)
    g(); touch()
    tc, tt, tb = t(g), t(touch), t(lambda: (touch(), g()))
    print(f"{name:14s} cold {tc:6.1f} us | touch alone {tt:5.1f} us | touch + gemm {tb:6.1f} us")
