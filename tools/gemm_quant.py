#!/usr/bin/env python3
"""What does tile quantisation cost?  Time vs M for the N = 384 / 768 shapes: 50688 rows are 2.32 rounds of tiles; compare
with row counts that make exactly 2 and exactly 3 rounds (cold caches, bf16 store epilogue)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda"); BF = torch.bfloat16
big = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
def t(fn):
    best = 1e9
    fn()
    for _ in range(6):
        big.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
FORCE = os.environ.get("DEVIT_GEMM_FORCE", "")
for name, N, K, bkm, tile, slots in (("S fc2 / fc1 dgrad", 384, 1536, 0, 128, 512), ("S qkv dgrad", 384, 1152, 1, 128, 512),
                                     ("S proj", 384, 384, 0, 128, 512), ("T proj", 768, 768, 0, 128, 512), ("T fc2", 768, 3072, 0, 256, 256)):
    nt = N // tile
    for rounds_m in (2 * slots // nt * tile, 50688, 3 * slots // nt * tile):
        M = rounds_m // 256 * 256
        a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn((K, N) if bkm else (N, K), device=dev) * .02).to(BF)
        out = torch.empty(M, N, dtype=BF, device=dev)
        us = t(lambda: ops.gemm(a, K, 0, w, N if bkm else K, bkm, M, N, K, kind=0, out=out, ldc=N))
        tiles = (M // tile) * nt
        print(f"{name:18s} M={M:6d} tiles={tiles:5d} rounds={tiles/slots:5.2f}  {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF  us/round-equivalent={us/(tiles/slots):6.1f}", flush=True)
