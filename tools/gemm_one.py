#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc). usage: gemm_one.py M N K kind a_km b_km reps"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = int(sys.argv[4]); a_km = int(sys.argv[5]); b_km = int(sys.argv[6]); reps = int(sys.argv[7])
dev = torch.device("cuda")
a = torch.randn((K, M) if a_km else (M, K), device=dev).to(torch.bfloat16)
b = (torch.randn((K, N) if b_km else (N, K), device=dev) * 0.02).to(torch.bfloat16)
out = torch.zeros((M, N), dtype=torch.float32 if kind in (L.EPI_ATOMIC_F32, L.EPI_STORE_F32) else torch.bfloat16, device=dev)
for _ in range(reps):
    ops.gemm(a, a.stride(0), a_km, b, b.stride(0), b_km, M, N, K, kind=kind, out=out, ldc=N, split_k=(16 if kind == L.EPI_ATOMIC_F32 else 1))
torch.cuda.synchronize()
