#!/usr/bin/env python3
"""Phase timeline of the attention FORWARD from in-kernel stamps (library built with -DDEVIT_ATTN_STAMP), teacher (H = 12)
and student (H = 6) at B = 256, cold."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops
from devit_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda"); B, N = 256, 198
flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
for H in (6, 12):
    D = H * 64; M = B * N
    qkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev, extra=128); qkv[:M] = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
    out = ops.rows_alloc(M, D, torch.bfloat16, dev); lse = torch.empty(B, H, N, device=dev)
    stamps = torch.zeros(B * H * 8, dtype=torch.int64, device=dev)
    for rep in range(3):
        flush.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("devit_attn_fwd", ptr(qkv), ptr(out), ptr(lse), ptr(stamps), B, N, H, 64, 0.125, 0, stream_ptr())
        e1.record(); torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 8).astype(np.float64)
    rt0, c0, c2, c3, c4, rt1, c6, c7 = (s[:, i] for i in range(8))
    t0 = rt0.min(); start, end = (rt0 - t0) / 100.0, (rt1 - t0) / 100.0
    life = end - start
    print(f"H = {H}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us; {len(s)} workgroups; lifetime median {np.median(life):.1f} us (p10 {np.percentile(life, 10):.1f}, p90 {np.percentile(life, 90):.1f})")
    print(f"  cycles (wave 0, median): loads issued {np.median(c6 - c0):.0f}  landed +{np.median(c7 - c6):.0f}  all waves' landed +{np.median(c2 - c7):.0f}"
          f"  compute + store issue {np.median(c3 - c2):.0f}  store drain {np.median(c4 - c3):.0f}  total {np.median(c4 - c0):.0f}")
    for t in (10, 30, 50):
        print(f"  resident workgroups at {t} us: {int(((start <= t) & (end > t)).sum())}")
