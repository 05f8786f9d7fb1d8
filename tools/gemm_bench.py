#!/usr/bin/env python3
"""Per-shape timing of the GEMM templates on the shapes of one DEKD step (MI355X). Random data."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L

dev = torch.device("cuda")
M = 50688
BF = torch.bfloat16
def rnd(*s, dt=BF, std=1.0): return (torch.randn(*s, device=dev) * std).to(dt)

COLD = os.environ.get("COLD", "0") == "1"     # flush the 256 MB Infinity Cache before every timed launch: in the step
_big = torch.empty(320 << 20, dtype=torch.uint8, device=dev) if COLD else None   # the operands come from HBM
def timeit(fn, reps=20):
    if COLD:
        fn(); best = 1e9
        for _ in range(6):
            _big.zero_(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3)
        return best
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

res = []
def report(name, flops, t):
    res.append((name, flops / t / 1e12, t * 1e6))
    print(f"{name:34s} {flops/t/1e12:8.1f} TF  {t*1e6:8.1f} us", flush=True)

for tag, D in (("S", 384), ("T", 768)):
    x = rnd(M, D); xh = rnd(M, 4 * D)
    wqkv, wproj, w1, w2 = rnd(3 * D, D, std=.02), rnd(D, D, std=.02), rnd(4 * D, D, std=.02), rnd(D, 4 * D, std=.02)
    bias3, bias1, biasd = rnd(3 * D, dt=torch.float32), rnd(4 * D, dt=torch.float32), rnd(D, dt=torch.float32)
    o3 = torch.empty(M, 3 * D, dtype=BF, device=dev); o4 = torch.empty(M, 4 * D, dtype=BF, device=dev); o4b = torch.empty_like(o4)
    res32 = rnd(M, D, dt=torch.float32); out32 = torch.empty_like(res32)
    od = torch.empty(M, D, dtype=BF, device=dev)
    report(f"{tag} qkv  NT store      N={3*D} K={D}", 2.0 * M * 3 * D * D, timeit(lambda: ops.gemm(x, D, 0, wqkv, D, 0, M, 3 * D, D, kind=L.EPI_STORE_BF16, out=o3, ldc=3 * D, bias=bias3)))
    report(f"{tag} proj NT residual   N={D} K={D}", 2.0 * M * D * D, timeit(lambda: ops.gemm(x, D, 0, wproj, D, 0, M, D, D, kind=L.EPI_RESIDUAL_F32, out=out32, ldc=D, bias=biasd, res=res32)))
    report(f"{tag} fc1  NT gelu(+pre) N={4*D} K={D}", 2.0 * M * 4 * D * D, timeit(lambda: ops.gemm(x, D, 0, w1, D, 0, M, 4 * D, D, kind=L.EPI_GELU_BF16, out=o4, ldc=4 * D, bias=bias1, aux=o4b if tag == "S" else None)))
    report(f"{tag} fc2  NT residual   N={D} K={4*D}", 2.0 * M * 4 * D * D, timeit(lambda: ops.gemm(xh, 4 * D, 0, w2, 4 * D, 0, M, D, 4 * D, kind=L.EPI_RESIDUAL_F32, out=out32, ldc=D, bias=biasd, res=res32)))
    if tag == "S":
        report("S fc2 dgrad dgelu     N=1536 K=384", 2.0 * M * 4 * D * D, timeit(lambda: ops.gemm(x, D, 0, w2, 4 * D, 1, M, 4 * D, D, kind=L.EPI_DGELU_BF16, out=o4, ldc=4 * D, aux_in=o4b)))
        report("S fc1 dgrad store     N=384 K=1536", 2.0 * M * 4 * D * D, timeit(lambda: ops.gemm(xh, 4 * D, 0, w1, D, 1, M, D, 4 * D, kind=L.EPI_STORE_BF16, out=od, ldc=D)))
        report("S qkv dgrad store     N=384 K=1152", 2.0 * M * 3 * D * D, timeit(lambda: ops.gemm(o3, 3 * D, 0, wqkv, D, 1, M, D, 3 * D, kind=L.EPI_STORE_BF16, out=od, ldc=D)))
        gw = torch.zeros(4 * D, D, device=dev)
        for sk in (7, 14, 28):
            report(f"S fc1 wgrad split_k={sk:2d}  [1536,384]", 2.0 * M * 4 * D * D, timeit(lambda: ops.gemm(xh, 4 * D, 1, x, D, 1, 4 * D, D, M, kind=L.EPI_ATOMIC_F32, out=gw, ldc=D, split_k=sk)))
        gq = torch.zeros(D, D, device=dev)
        for sk in (28, 56, 113):
            report(f"S proj wgrad split_k={sk:3d} [384,384]", 2.0 * M * D * D, timeit(lambda: ops.gemm(x, D, 1, x, D, 1, D, D, M, kind=L.EPI_ATOMIC_F32, out=gq, ldc=D, split_k=sk)))
        bgk = torch.zeros(3 * D, device=dev); bg1 = torch.zeros(4 * D, device=dev)
        report("S fc1 wgrad split_k=14 + bias grad (fused)", 2.0 * M * 4 * D * D, timeit(lambda: ops.gemm(xh, 4 * D, 1, x, D, 1, 4 * D, D, M, kind=L.EPI_ATOMIC_F32, out=gw, ldc=D, split_k=14, aux=bg1)))
        report("S fc1 bias grad alone (colsum pass)", 2.0 * M * 4 * D * D, timeit(lambda: ops.colsum(xh, M, 4 * D, bg1, accumulate=True)))
        gk = torch.zeros(3 * D, D, device=dev)
        report("S qkv wgrad split_k=18 + bias grad (fused)", 2.0 * M * 3 * D * D, timeit(lambda: ops.gemm(o3, 3 * D, 1, x, D, 1, 3 * D, D, M, kind=L.EPI_ATOMIC_F32, out=gk, ldc=D, split_k=18, aux=bgk)))
        for sk in (9, 18, 37):
            report(f"S qkv wgrad split_k={sk:3d} [1152,384]", 2.0 * M * 3 * D * D, timeit(lambda: ops.gemm(o3, 3 * D, 1, x, D, 1, 3 * D, D, M, kind=L.EPI_ATOMIC_F32, out=gk, ldc=D, split_k=sk)))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/gemm_bench.json", "w"))
