#!/usr/bin/env python3
"""Phase timeline of the 4-wave attention backward from in-kernel stamps (library built with -DDEVIT_ATTN_STAMP:
tools/build_variant.sh attnstamp "-DDEVIT_ATTN_STAMP"; DEVIT_LIB_PATH=tools/_diag/libdevit_attnstamp.so).
Per workgroup: entry (realtime + cycles), prologue done, main loop done, end (cycles + realtime)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops
from devit_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda"); B, N, H = 256, 198, int(os.environ.get("H", 6))
D = H * 64; M = B * N
qkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev, extra=128); qkv[:M] = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
out = ops.rows_alloc(M, D, torch.bfloat16, dev); lse = torch.empty(B, H, N, device=dev)
dout = ops.rows_alloc(M, D, torch.bfloat16, dev); dout[:M] = torch.randn(M, D, device=dev).to(torch.bfloat16)
dqkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev)
call("devit_attn_fwd", ptr(qkv), ptr(out), ptr(lse), None, B, N, H, 64, 0.125, 0, stream_ptr())
stamps = torch.zeros(B * H * 8, dtype=torch.int64, device=dev)
flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
for rep in range(3):
    flush.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call("devit_attn_bwd", ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(stamps), None, ptr(dqkv), B, N, H, 64, 0.125, stream_ptr())
    e1.record(); torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 8).astype(np.float64)
rt0, c0, c2, c3, c4, rt1 = s[:, 0], s[:, 1], s[:, 2], s[:, 3], s[:, 4], s[:, 5]
c6, c7 = s[:, 6], s[:, 7]
print(f"prologue split (cycles, median): loads issued {np.median(c6 - c0):.0f}  landed +{np.median(c7 - c6):.0f}  delta / zero / barrier +{np.median(c2 - c7):.0f}")
first = np.argsort(rt0)[:512]
late = np.argsort(rt0)[512:]
print(f"  first round (512 simultaneous starts): landed +{np.median((c7 - c6)[first]):.0f};  later rounds: landed +{np.median((c7 - c6)[late]):.0f}, main loop {np.median((c3 - c2)[late]):.0f}")
t0 = rt0.min()
start, end = (rt0 - t0) / 100.0, (rt1 - t0) / 100.0          # us (100 MHz realtime counter)
print(f"kernel {e0.elapsed_time(e1) * 1e3:.1f} us by events; first start -> last end {end.max():.1f} us; {len(s)} workgroups")
life = end - start
print(f"workgroup lifetime us: median {np.median(life):.1f}  p10 {np.percentile(life, 10):.1f}  p90 {np.percentile(life, 90):.1f}")
cyc = c4 - c0
print(f"cycles: prologue {np.median(c2 - c0):.0f}  main loop {np.median(c3 - c2):.0f}  epilogue {np.median(c4 - c3):.0f}  total {np.median(cyc):.0f}"
      f"  (clock ~ {np.median(cyc / life) / 1e3:.2f} GHz)")
for t in (10, 30, 60, 90, 120):
    print(f"  resident workgroups at {t:3d} us: {int(((start <= t) & (end > t)).sum())}")
order = np.argsort(start)
print("start times of the 1st / 512th / 513th / 1024th / 1536th workgroup:", [round(float(start[order[i]]), 1) for i in (0, 511, 512, 1023, len(s) - 1)])
