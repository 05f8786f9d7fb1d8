#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda")
def bench(M, N, K, lda, ldb, reps=10):
    a = torch.randn(M, lda, device=dev).to(torch.bfloat16); b = (torch.randn(N, ldb, device=dev) * .02).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fn = lambda: ops.gemm(a, lda, 0, b, ldb, 0, M, N, K, kind=0, out=out, ldc=N)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print(f"{M}x{N}x{K} lda {lda} ldb {ldb}: {2.0*M*N*K/t/1e12:7.1f} TF  {t*1e6:8.1f} us", flush=True)
for rep in range(2):
    for lda, ldb in ((768, 768), (832, 768), (768, 832), (832, 832), (1024, 1024), (776, 776), (800, 800)):
        bench(50688, 2304, 768, lda, ldb)
for lda in (384, 448, 392):
    bench(50688, 1152, 384, lda, lda)
