#!/usr/bin/env python3
"""Student fc1 (M=50688, N=1536, K=384) under different epilogues / tile configs: where do its 129 us go?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda"); M, D = 50688, 384; BF = torch.bfloat16
rnd = lambda *s, dt=BF, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
x, w1, b1 = rnd(M, D), rnd(4 * D, D, std=.02), rnd(4 * D, dt=torch.float32)
w2 = rnd(D, 4 * D, std=.02)
o4, o4b = torch.empty(M, 4 * D, dtype=BF, device=dev), torch.empty(M, 4 * D, dtype=BF, device=dev)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
fl = 2.0 * M * 4 * D * D
cases = {
 "store bf16": lambda: ops.gemm(x, D, 0, w1, D, 0, M, 4 * D, D, kind=L.EPI_STORE_BF16, out=o4, ldc=4 * D, bias=b1),
 "gelu": lambda: ops.gemm(x, D, 0, w1, D, 0, M, 4 * D, D, kind=L.EPI_GELU_BF16, out=o4, ldc=4 * D, bias=b1),
 "gelu + pre": lambda: ops.gemm(x, D, 0, w1, D, 0, M, 4 * D, D, kind=L.EPI_GELU_BF16, out=o4, ldc=4 * D, bias=b1, aux=o4b),
 "dgelu dgrad": lambda: ops.gemm(x, D, 0, w2, 4 * D, 1, M, 4 * D, D, kind=L.EPI_DGELU_BF16, out=o4, ldc=4 * D, aux_in=o4b),
}
for name, fn in cases.items():
    t = timeit(fn)
    print(f"{os.environ.get('DEVIT_GEMM_FORCE','auto'):>4s} {name:14s} {t:7.1f} us  {fl/t/1e6:7.1f} TF", flush=True)
