#!/usr/bin/env python3
"""In-kernel timeline of the four-wave GEMM (library built with -DDEVIT_GEMM4_STAMP: tools/build_variant.sh g4stamp "-DDEVIT_GEMM4_STAMP").
Per wave: tiles, cycles in the K loop / epilogue, and inside the loop: entry reads, phase 1 (issue of 64 MFMAs + reads + A requests),
the middle (counted vmcnt + lgkmcnt + barrier), phase 2 (64 MFMAs + reads + B requests + end wait).
usage: DEVIT_LIB_PATH=tools/_diag/libdevit_g4stamp.so gemm4_stamps.py [N K kind]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
M = 50688; N = int(sys.argv[1]) if len(sys.argv) > 1 else 2304; K = int(sys.argv[2]) if len(sys.argv) > 2 else 768
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda")
a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
assert kind in (L.EPI_STORE_BF16, L.EPI_GELU_BF16, L.EPI_RESIDUAL_F32, L.EPI_STORE_F32)
f32_out = kind in (L.EPI_RESIDUAL_F32, L.EPI_STORE_F32)
out = torch.empty(M, N, dtype=torch.float32 if f32_out else torch.bfloat16, device=dev)
res = torch.randn(M, N, device=dev) if kind == L.EPI_RESIDUAL_F32 else None
dbg = torch.zeros(256 * 4 * 16, dtype=torch.int64, device=dev)
big = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
fn = lambda: ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, pos=dbg.view(torch.float32), bias=torch.zeros(N, device=dev), res=res)
for _ in range(3): fn()
big.zero_(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
print(f"N={N} K={K} kind={kind}: kernel by events (cold, stamped build): {e0.elapsed_time(e1) * 1e3:.1f} us")
d = dbg.view(256 * 4, 16).cpu().double()
d = d[d[:, 0] > 0]
nk = K // 64
tiles = d[:, 0]
med = lambda x: float(x.median())
print(f"waves {len(d)}, tiles per wave {int(tiles.min())}..{int(tiles.max())}, K-steps per tile {nk}")
print(f"per tile (median over waves of per-wave means): K loop {med(d[:, 1] / tiles):8.0f}   epilogue {med(d[:, 2] / tiles):8.0f}   first tile's loop {med(d[:, 7]):8.0f}")
print(f"per K-step: phase 1 {med(d[:, 4] / tiles / nk):7.0f}   middle (waits + barrier) {med(d[:, 5] / tiles / nk):7.0f}   phase 2 {med(d[:, 6] / tiles / nk):7.0f}   "
      f"sum {med((d[:, 4] + d[:, 5] + d[:, 6]) / tiles / nk):7.0f}   (MFMA-bound 2048);  entry reads per tile {med(d[:, 3] / tiles):6.0f}")
dur = d[:, 9] - d[:, 8]; rdur = (d[:, 11] - d[:, 10]) / 100.0
print(f"per wave entry->exit: median {med(dur):.0f} cycles = {med(rdur):.1f} us (clock {med(dur / rdur):.0f} MHz); exit spread {(d[:, 11].max() - d[:, 11].min()) / 100.0:.1f} us")
