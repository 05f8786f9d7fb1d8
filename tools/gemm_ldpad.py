#!/usr/bin/env python3
"""Does the leading dimension of the streamed operand matter (L2 / HBM channel interleave)?  The step's forward GEMM shapes with the
activation operand (and the output) at ld = K (+ PAD elements): rows that are a multiple of 512 B apart put a stage's 256 row pieces on few
channels.  COLD=1 flushes the Infinity Cache before every launch.  usage: gemm_ldpad.py [pad ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L

dev = torch.device("cuda")
M = 50688
BF = torch.bfloat16
pads = [int(x) for x in sys.argv[1:]] or [0, 8, 32, 64, 72]
big = torch.empty(320 << 20, dtype=torch.uint8, device=dev)

def timeit(fn):
    fn(); best = []
    for _ in range(5):
        big.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3)
    return min(best)

shapes = [("T qkv  store", 2304, 768, L.EPI_STORE_BF16), ("T fc1  gelu", 3072, 768, L.EPI_GELU_BF16), ("T fc2  residual", 768, 3072, L.EPI_RESIDUAL_F32),
          ("T proj residual", 768, 768, L.EPI_RESIDUAL_F32), ("S qkv  store", 1152, 384, L.EPI_STORE_BF16), ("S fc1  gelu", 1536, 384, L.EPI_GELU_BF16),
          ("S fc2  residual", 384, 1536, L.EPI_RESIDUAL_F32)]
for name, N, K, kind in shapes:
    w = (torch.randn(N, K, device=dev) * 0.02).to(BF)
    bias = torch.randn(N, device=dev)
    row = []
    for pad in pads:
        lda = K + pad
        a = torch.randn(M, lda, device=dev).to(BF)
        f32 = kind == L.EPI_RESIDUAL_F32
        ldc = N + (pad if not f32 else pad // 2 * 2)
        out = torch.empty(M, ldc, dtype=torch.float32 if f32 else BF, device=dev)
        res = torch.randn(M, ldc, device=dev) if f32 else None
        t = timeit(lambda: ops.gemm(a, lda, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=ldc, bias=bias, res=res))
        row.append(f"pad {pad:3d}: {t:7.1f} us")
        del a, out, res
    print(f"{name:16s} N={N:5d} K={K:5d}   " + "   ".join(row), flush=True)
