#!/usr/bin/env python3
"""The student's N = 384 launches on the 128x128 kernels (DEVIT_GEMMFR=0) and on the full-row 256x384 kernel (DEVIT_GEMMFR=1): microseconds per
launch, warm (back to back) and cold (the Infinity Cache flushed before every timed launch: the state the step's launches run in), interleaved."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L

dev = torch.device("cuda"); M = 50688; BF = torch.bfloat16
def rnd(*s, dt=BF, std=1.0): return (torch.randn(*s, device=dev) * std).to(dt)
big = torch.empty(320 << 20, dtype=torch.uint8, device=dev)

def cold(fn, n=5):
    best = 1e9
    for _ in range(n):
        big.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best

def warm(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

D = 384
x, xh, x3 = rnd(M, D), rnd(M, 4 * D), rnd(M, 3 * D)
wproj, w1, w2, wqkv = rnd(D, D, std=.02), rnd(4 * D, D, std=.02), rnd(D, 4 * D, std=.02), rnd(3 * D, D, std=.02)
bias, bias3 = rnd(D, dt=torch.float32), rnd(3 * D, dt=torch.float32)
res32 = rnd(M, D, dt=torch.float32); out32 = torch.empty_like(res32)
od = torch.empty(M, D, dtype=BF, device=dev); o3 = torch.empty(M, 3 * D, dtype=BF, device=dev)
wprojT, w2T = wproj.t().contiguous(), w2.t().contiguous()       # k-major copies [K][N] of the forward weights
# (name, flops, 128x128 kernels, full-row kernel)
shapes = [
    ("proj  resid   K=384 ", 2.0 * M * D * D, lambda: ops.gemm(x, D, 0, wproj, D, 0, M, D, D, kind=L.EPI_RESIDUAL_F32, out=out32, ldc=D, bias=bias, res=res32),
     lambda: ops.gemm(x, D, 0, wprojT, D, 1, M, D, D, kind=L.EPI_RESIDUAL_F32, out=out32, ldc=D, bias=bias, res=res32)),
    ("fc2   resid   K=1536", 2.0 * M * 4 * D * D, lambda: ops.gemm(xh, 4 * D, 0, w2, 4 * D, 0, M, D, 4 * D, kind=L.EPI_RESIDUAL_F32, out=out32, ldc=D, bias=bias, res=res32),
     lambda: ops.gemm(xh, 4 * D, 0, w2T, D, 1, M, D, 4 * D, kind=L.EPI_RESIDUAL_F32, out=out32, ldc=D, bias=bias, res=res32)),
    ("fc1 dgrad km  K=1536", 2.0 * M * 4 * D * D, lambda: ops.gemm(xh, 4 * D, 0, w1, D, 1, M, D, 4 * D, kind=L.EPI_STORE_BF16, out=od, ldc=D), None),
    ("qkv dgrad km  K=1152", 2.0 * M * 3 * D * D, lambda: ops.gemm(x3, 3 * D, 0, wqkv, D, 1, M, D, 3 * D, kind=L.EPI_STORE_BF16, out=od, ldc=D), None),
    ("proj dgrad km K=384 ", 2.0 * M * D * D, lambda: ops.gemm(x, D, 0, wproj, D, 1, M, D, D, kind=L.EPI_STORE_BF16, out=od, ldc=D), None),
]
print(f"{'shape':22s} {'128^2 warm':>10s} {'FR warm':>8s} {'128^2 cold':>10s} {'FR cold':>8s}   TF/s cold (old -> FR)")
for name, fl, fn, fn_fr in shapes:
    r = {}
    for rep in range(2):
        for flag in ("0", "1"):
            os.environ["DEVIT_GEMMFR"] = flag
            f = fn if flag == "0" or fn_fr is None else fn_fr
            r[flag] = (min(r.get(flag, (1e9, 1e9))[0], warm(f)), min(r.get(flag, (1e9, 1e9))[1], cold(f)))
    print(f"{name:22s} {r['0'][0]:10.1f} {r['1'][0]:8.1f} {r['0'][1]:10.1f} {r['1'][1]:8.1f}   {fl / r['0'][1] / 1e6:6.0f} -> {fl / r['1'][1] / 1e6:6.0f}", flush=True)
