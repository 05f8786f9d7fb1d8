#!/usr/bin/env python3
"""Cold-cache timing of the launches that run on the 256x256 ping-pong GEMM (M = 50688): teacher qkv / fc1 / fc2, student
fc1 (GELU + stored pre-activation) and the student's fc2 dgrad (dGELU epilogue).  One library per process (DEVIT_LIB_PATH);
tools/gpu_r02i.sh interleaves the variants.  Prints the median of 7 launches, Infinity Cache flushed before each."""
import os, sys, statistics as st
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda"); M, BF = 50688, torch.bfloat16
rnd = lambda *s, dt=BF, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
tag = os.path.basename(os.environ.get("DEVIT_LIB_PATH", "default")).replace("libdevit_", "").replace(".so", "")
D = 768
x, xh = rnd(M, D), rnd(M, 4 * D)
wqkv, w1, w2 = rnd(3 * D, D, std=.02), rnd(4 * D, D, std=.02), rnd(D, 4 * D, std=.02)
b3, b1, bd = rnd(3 * D, dt=torch.float32), rnd(4 * D, dt=torch.float32), rnd(D, dt=torch.float32)
o3 = torch.empty(M, 3 * D, dtype=BF, device=dev); o4 = torch.empty(M, 4 * D, dtype=BF, device=dev)
res32 = rnd(M, D, dt=torch.float32); out32 = torch.empty_like(res32)
Ds = 384
xs = rnd(M, Ds); w1s = rnd(4 * Ds, Ds, std=.02); w2s = rnd(Ds, 4 * Ds, std=.02); b1s = rnd(4 * Ds, dt=torch.float32)
o4s = torch.empty(M, 4 * Ds, dtype=BF, device=dev); o4p = rnd(M, 4 * Ds)
shapes = [
    ("T qkv ", 2.0 * M * 3 * D * D, lambda: ops.gemm(x, D, 0, wqkv, D, 0, M, 3 * D, D, kind=L.EPI_STORE_BF16, out=o3, ldc=3 * D, bias=b3)),
    ("T fc1 ", 2.0 * M * 4 * D * D, lambda: ops.gemm(x, D, 0, w1, D, 0, M, 4 * D, D, kind=L.EPI_GELU_BF16, out=o4, ldc=4 * D, bias=b1)),
    ("T fc2 ", 2.0 * M * 4 * D * D, lambda: ops.gemm(xh, 4 * D, 0, w2, 4 * D, 0, M, D, 4 * D, kind=L.EPI_RESIDUAL_F32, out=out32, ldc=D, bias=bd, res=res32)),
    ("S fc1 ", 2.0 * M * 4 * Ds * Ds, lambda: ops.gemm(xs, Ds, 0, w1s, Ds, 0, M, 4 * Ds, Ds, kind=L.EPI_GELU_BF16, out=o4s, ldc=4 * Ds, bias=b1s, aux=o4p)),
    ("S dgelu", 2.0 * M * 4 * Ds * Ds, lambda: ops.gemm(xs, Ds, 0, w2s, 4 * Ds, 1, M, 4 * Ds, Ds, kind=L.EPI_DGELU_BF16, out=o4s, ldc=4 * Ds, aux_in=o4p)),
]
out = []
for name, flops, fn in shapes:
    fn(); fn(); ts = []
    for _ in range(7):
        flush.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    out.append(f"{name} {st.median(ts):6.1f}")
print(f"{tag:8s} " + " | ".join(out) + "   (us, median of 7, cold)", flush=True)
