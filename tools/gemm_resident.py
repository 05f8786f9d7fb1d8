#!/usr/bin/env python3
"""Is the forward GEMM bound by how fast its A operand arrives?  Same FLOPs, tiles and epilogue as the teacher shapes at
M = 50688, but every 256-row m-tile reads the SAME 256 rows of A (batch = 198 with a batch stride of 0): A is then
L2-resident for the whole launch.  Compared with the real launch (A streamed once from HBM / the Infinity Cache)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda"); BF = torch.bfloat16; M = 50688
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def t(fn, cold):
    fn(); best = 1e9
    for _ in range(6):
        if cold: big.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best
for name, N, K, kind in (("T qkv", 2304, 768, L.EPI_STORE_BF16), ("T fc1", 3072, 768, L.EPI_GELU_BF16), ("T fc2", 768, 3072, L.EPI_STORE_BF16),
                         ("S fc1", 1536, 384, L.EPI_GELU_BF16)):
    a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.02).to(BF)
    bias = torch.randn(N, device=dev); out = torch.empty(M, N, dtype=BF, device=dev)
    fl = 2.0 * M * N * K
    real = lambda: ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, bias=bias)
    res = lambda: ops.gemm(a, K, 0, w, K, 0, 256, N, K, kind=kind, out=out, ldc=N, bias=bias, batch=M // 256, a_bs=0, b_bs=0, out_bs=256 * N)
    for tag, fn in (("streamed A", real), ("resident A", res)):
        for cold in (0, 1):
            dt = t(fn, cold)
            print(f"{name} N={N} K={K} {tag:11s} cold={cold}: {fl / dt / 1e12:7.1f} TF {dt * 1e6:7.1f} us", flush=True)
