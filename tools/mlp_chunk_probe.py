#!/usr/bin/env python3
"""Would row-chunking the teacher's MLP keep h in the Infinity Cache?  fc1 (+GELU) -> fc2 (fp32 residual) over all 50688 rows against the same pair
over row chunks of 85 / 85 / 28 m-tiles (the same number of tile rounds on 256 CUs: 4 + 4 + 2 and 1 + 1 + 1) and of 64 / 64 / 64 / 6; us per pair, other
traffic flushed in between (a 320 MB fill stands for the rest of the step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda"); BF = torch.bfloat16
M, D, Hd = 50688, 768, 3072
x = torch.randn(M, D, device=dev).to(BF); w1 = (torch.randn(Hd, D, device=dev) * 0.02).to(BF); w2 = (torch.randn(D, Hd, device=dev) * 0.02).to(BF)
b1 = torch.zeros(Hd, device=dev); b2 = torch.zeros(D, device=dev)
h = torch.empty(M, Hd, dtype=BF, device=dev); res = torch.randn(M, D, device=dev); out = torch.empty(M, D, device=dev)
big = torch.empty(320 << 20, dtype=torch.uint8, device=dev)

def pair(r0, r1):
    n = r1 - r0
    ops.gemm(x[r0:r1], D, 0, w1, D, 0, n, Hd, D, kind=L.EPI_GELU_BF16, out=h[r0:r1], ldc=Hd, bias=b1)
    ops.gemm(h[r0:r1], Hd, 0, w2, Hd, 0, n, D, Hd, kind=L.EPI_RESIDUAL_F32, out=out[r0:r1], ldc=D, bias=b2, res=res[r0:r1])

def run(chunks):
    best = 1e9
    for _ in range(5):
        big.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = 0
        for c in chunks:
            pair(r, r + c * 256); r += c * 256
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
for name, ch in (("one launch pair, 198 m-tiles", [198]), ("85 / 85 / 28", [85, 85, 28]), ("64 / 64 / 64 / 6", [64, 64, 64, 6]), ("43 x 4 + 26", [43, 43, 43, 43, 26]), ("99 / 99", [99, 99])):
    print(f"{name:32s} {run(ch):8.1f} us", flush=True)
