#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda")
def bench(M, N, K, kind=0, reps=10, note=""):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = (torch.randn(N, K, device=dev) * .02).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.float32 if kind == 6 else torch.bfloat16, device=dev)
    fn = lambda: ops.gemm(a, K, 0, b, K, 0, M, N, K, kind=kind, out=out, ldc=N)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    wgs = (M // 256) * (N // 256)
    print(f"{M:6d}x{N:5d}x{K:5d} kind {kind} WGs {wgs:5d} ({wgs/256:5.2f} rounds)  {2.0*M*N*K/t/1e12:7.1f} TF  {t*1e6:8.1f} us {note}", flush=True)
for M in (8192, 16384, 32768, 50688, 65536):
    bench(M, 2304, 768)
for K in (512, 768, 1536, 3072, 6144):
    bench(8192, 2048, K)
for K in (768, 1536, 3072):
    bench(50688, 2304, K)
bench(50688, 2304, 768, kind=6, note="f32 out")
bench(65536, 2048, 768)
bench(65536, 4096, 768)
bench(65536, 4096, 4096)
