#!/usr/bin/env python3
"""TFLOP/s of one GEMM shape, warm.  usage: gemm_shape.py M N K kind [b_km]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
M, N, K, kind = (int(v) for v in sys.argv[1:5]); b_km = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda")
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
b = (torch.randn((K, N) if b_km else (N, K), device=dev) * 0.02).to(torch.bfloat16)
f32 = kind in (L.EPI_STORE_F32, L.EPI_RESIDUAL_F32)
out = torch.zeros((M, N), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
kw = dict(bias=torch.randn(N, device=dev))
if kind == L.EPI_RESIDUAL_F32: kw["res"] = torch.randn(M, N, device=dev)
if kind == L.EPI_DGELU_BF16: kw = dict(aux_in=torch.randn(M, N, device=dev).to(torch.bfloat16))
fn = lambda: ops.gemm(a, K, 0, b, b.stride(0), b_km, M, N, K, kind=kind, out=out, ldc=N, **kw)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e-3
print(f"{os.environ.get('DEVIT_LIB_PATH', 'default')[-24:]:24s} M={M} N={N} K={K} kind={kind}: {2.0 * M * N * K / t / 1e12:7.1f} TF {t * 1e6:7.1f} us", flush=True)
