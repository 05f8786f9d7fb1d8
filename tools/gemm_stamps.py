#!/usr/bin/env python3
"""In-kernel timeline of one K-step of the ping-pong GEMM (diagnostic build -DDEVIT_GEMM_STAMP): s_memtime at the edges
of its four barrier intervals, per wave of each workgroup; prints the median durations for the leading (wm = 0) and the
lagging (wm = 1) wave group.  usage: DEVIT_LIB_PATH=tools/_diag/libdevit_stamp.so gemm_stamps.py [N K]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
M = 50688; N = int(sys.argv[1]) if len(sys.argv) > 1 else 2304; K = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev = torch.device("cuda")
a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
dbg = torch.zeros(256 * 8 * 12, dtype=torch.int64, device=dev)
fn = lambda: ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=L.EPI_STORE_BF16, out=out, ldc=N, pos=dbg.view(torch.float32))
for _ in range(5): fn()
torch.cuda.synchronize()
d = dbg.view(256, 8, 12).cpu().double()
names = ["reads issued", "DMA wait", "lgkmcnt(0)", "barrier", "MFMAs issued", "barrier"]
for grp, sl in (("wm=0 (leading)", slice(0, 4)), ("wm=1 (lagging)", slice(4, 8))):
    x = d[:, sl, :].reshape(-1, 12)
    x = x[x[:, 0] > 0]
    t0 = x[:, 0:1]
    rel = (x - t0)
    med = rel.median(0).values
    print(grp, "cycles since 'reads issued' of kk=0 (median over", x.shape[0], "waves):")
    prev = 0.0
    for q in range(12):
        print(f"   kk={q // 6} {names[q % 6]:14s} {med[q]:8.0f}  (+{med[q] - prev:6.0f})")
        prev = float(med[q])
# offset between the two groups on the same workgroup
off = (d[:, 4:8, 0] - d[:, 0:4, 0]).reshape(-1)
print("lagging group's kk=0 'reads issued' minus leading group's:", float(off.median()))
