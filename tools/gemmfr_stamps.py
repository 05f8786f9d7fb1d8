#!/usr/bin/env python3
"""In-kernel timeline of the full-row GEMM (library built with -DDEVIT_GEMMFR_STAMP: tools/build_variant.sh frstamp "-DDEVIT_GEMMFR_STAMP").
Per wave: prologue (first two stages requested -> landed), K loop and epilogue per tile, and inside the loop the K-step (stamp to stamp in
front of each step's barrier) and the share of it spent in that barrier's waits (vmcnt(0) = the next stage landed, lgkmcnt(0), s_barrier).
usage: DEVIT_LIB_PATH=tools/_diag/libdevit_frstamp.so gemmfr_stamps.py [K resid]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DEVIT_GEMMFR"] = "1"
from devit_amd import ops, _lib as L
M = 50688; N = 384; K = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
resid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda")
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn((K, N), device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.float32 if resid else torch.bfloat16, device=dev)
res = torch.randn(M, N, device=dev) if resid else None
dbg = torch.zeros(256 * 4 * 8, dtype=torch.int64, device=dev)
big = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
fn = lambda: ops.gemm(a, K, 0, w, N, 1, M, N, K, kind=L.EPI_RESIDUAL_F32 if resid else L.EPI_STORE_BF16, out=out, ldc=N,
                      pos=dbg.view(torch.float32), res=res)
for _ in range(3): fn()
big.zero_(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
print(f"K={K} {'fp32 residual' if resid else 'bf16 store'}: kernel by events (cold, stamped build): {e0.elapsed_time(e1) * 1e3:.1f} us")
d = dbg.view(256 * 4, 8).cpu().double()
d = d[d[:, 0] > 0]
nk = K // 64; tiles = d[:, 0]
med = lambda x: float(x.median())
print(f"waves {len(d)}, tiles per wave {int(tiles.min())}..{int(tiles.max())}, K-steps per tile {nk}")
print(f"prologue {med(d[:, 5]):7.0f}   K loop per tile {med(d[:, 1] / tiles):8.0f}   epilogue per tile {med(d[:, 2] / tiles):8.0f}   entry->exit {med(d[:, 7] - d[:, 6]):8.0f} cycles")
print(f"per K-step (entry .. last barrier) {med(d[:, 3] / tiles / nk):7.0f} (MFMA-bound 3072), of it in the two barriers' waits {med(d[:, 4] / tiles / nk):7.0f}")
print(f"exit spread over waves {float(d[:, 7].max() - d[:, 7].min()):.0f} cycles; entry spread {float(d[:, 6].max() - d[:, 6].min()):.0f}")
