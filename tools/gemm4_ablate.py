#!/usr/bin/env python3
"""Time the teacher's three 256x256 GEMM shapes (qkv store, fc1 GELU, fc2 residual K=3072) with whatever library DEVIT_LIB_PATH
names and DEVIT_GEMM4 selects; cold Infinity Cache; min of 6.  One line per shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda"); M = 50688; BF = torch.bfloat16
big = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
def t(fn):
    fn(); best = 1e9
    for _ in range(6):
        big.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
D = 768
x = torch.randn(M, D, device=dev).to(BF); xh = torch.randn(M, 4 * D, device=dev).to(BF)
wqkv = (torch.randn(3 * D, D, device=dev) * .02).to(BF); w1 = (torch.randn(4 * D, D, device=dev) * .02).to(BF); w2 = (torch.randn(D, 4 * D, device=dev) * .02).to(BF)
b3, b1, bd = torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev), torch.randn(D, device=dev)
o3 = torch.empty(M, 3 * D, dtype=BF, device=dev); o4 = torch.empty(M, 4 * D, dtype=BF, device=dev)
r32 = torch.randn(M, D, device=dev); o32 = torch.empty_like(r32)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
a = t(lambda: ops.gemm(x, D, 0, wqkv, D, 0, M, 3 * D, D, kind=L.EPI_STORE_BF16, out=o3, ldc=3 * D, bias=b3))
b = t(lambda: ops.gemm(x, D, 0, w1, D, 0, M, 4 * D, D, kind=L.EPI_GELU_BF16, out=o4, ldc=4 * D, bias=b1))
c = t(lambda: ops.gemm(xh, 4 * D, 0, w2, 4 * D, 0, M, D, 4 * D, kind=L.EPI_RESIDUAL_F32, out=o32, ldc=D, bias=bd, res=r32))
print(f"{tag:28s} qkv {a:7.1f} us   fc1 {b:7.1f} us   fc2 {c:7.1f} us", flush=True)
