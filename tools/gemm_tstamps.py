#!/usr/bin/env python3
"""Tile-level in-kernel timeline of the ping-pong GEMM (diagnostic build -DDEVIT_GEMM_TSTAMP): per workgroup and wave,
s_memtime at K-loop start / K-loop end / epilogue end of its first eight tiles and at the end of every K-step of its
second tile.  usage: DEVIT_LIB_PATH=tools/_diag/libdevit_tstamp.so gemm_tstamps.py [N K kind]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
M = 50688; N = int(sys.argv[1]) if len(sys.argv) > 1 else 2304; K = int(sys.argv[2]) if len(sys.argv) > 2 else 768
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda")
a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
# the stamp build borrows ep.pos as its debug buffer: only kinds that never read pos (not PATCH), forward layouts only
assert kind in (L.EPI_STORE_BF16, L.EPI_GELU_BF16, L.EPI_RESIDUAL_F32, L.EPI_STORE_F32), f"kind {kind} is not supported by this tool"
f32_out = kind in (L.EPI_RESIDUAL_F32, L.EPI_STORE_F32)           # the epilogue writes M x N floats: size the buffer for them
out = torch.empty(M, N, dtype=torch.float32 if f32_out else torch.bfloat16, device=dev)
res = torch.randn(M, N, device=dev) if kind == L.EPI_RESIDUAL_F32 else None
dbg = torch.zeros(256 * 8 * 48, dtype=torch.int64, device=dev)
fn = lambda: ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, pos=dbg.view(torch.float32), bias=torch.zeros(N, device=dev), res=res)
for _ in range(5): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
print(f"kernel by events: {e0.elapsed_time(e1) * 1e3:.1f} us")
d = dbg.view(256, 8, 48).cpu().double()
for grp, sl in (("wm=0", slice(0, 4)), ("wm=1", slice(4, 8))):
    x = d[:, sl, :].reshape(-1, 48)
    t0 = x[:, 0:1]
    tiles = (x[:, :24] - t0).view(-1, 8, 3)
    print(grp, "per tile [K-loop start, K-loop end, epilogue end] (median cycles since the first tile's K-loop start):")
    for i in range(8):
        col = tiles[:, i, :]
        ok = x[:, i * 3 + 2] > 0
        if ok.sum() == 0: break
        m = col[ok].median(0).values
        print(f"   tile {i}: start {m[0]:8.0f}  kloop {m[1] - m[0]:7.0f}  epilogue {m[2] - m[1]:7.0f}   ({int(ok.sum())} waves)")
    ks = x[:, 24:24 + K // 64]
    st = x[:, 3:4]
    dk = torch.cat([ks[:, :1] - st, ks[:, 1:] - ks[:, :-1]], 1)
    print("   K-steps of tile 1 (median cycles each):", [int(v) for v in dk.median(0).values])

x = d.reshape(-1, 48)
dur = x[:, 41] - x[:, 40]; rdur = (x[:, 43] - x[:, 42]) / 100.0
print(f"per wave entry->exit: median {dur.median():.0f} cycles = {rdur.median():.1f} us (clock {float((dur / rdur).median()):.0f} MHz); "
      f"entry->ring primed {float((x[:, 44] - x[:, 40]).median()):.0f} cycles; tiles per workgroup {x[:, 45].min():.0f}..{x[:, 45].max():.0f}")
first_entry = x[:, 42].min(); last_exit = x[:, 43].max()
print(f"first wave entry -> last wave exit: {(last_exit - first_entry) / 100.0:.1f} us; entry spread {(x[:, 42].max() - first_entry) / 100.0:.1f} us; "
      f"exit spread {(last_exit - x[:, 43].min()) / 100.0:.1f} us")
