#!/usr/bin/env python3
"""Diagnostic: per-wave cycle split (prologue / K-loop / epilogue) of the 128x128 GEMM tile on the student shapes.
Needs tools/_diag/libdevit_hip_stamps.so (tools/build_stamps.sh); run with DEVIT_GEMM_TILE=1."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from devit_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "_diag", "libdevit_hip_stamps.so")
from devit_amd import ops
lib = L.load()
dev = torch.device("cuda")
def run(M, N, K, b_km, kind, label, aux=False, res=False):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    b = (torch.randn((K, N) if b_km else (N, K), device=dev) * 0.02).to(torch.bfloat16)
    f32 = kind in (2, 5, 6)
    out = torch.zeros((M, N), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    kw = {}
    if aux: kw["aux"] = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    if kind == 2: kw["res"] = torch.randn((M, N), device=dev)
    if kind == 4: kw["aux_in"] = torch.randn((M, N), device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    buf = (C.c_ulonglong * 8)()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(3):
        torch.cuda.synchronize(); ev0.record()
        ops.gemm(a, a.stride(0), False, b, b.stride(0), b_km, M, N, K, kind=kind, out=out, ldc=N, bias=bias, **kw)
        ev1.record(); torch.cuda.synchronize()
        lib.devit_debug_gemm_stamps(buf, 1)
    waves = (M // 128) * (N // 128) * 4
    nk = K // 64
    us = ev0.elapsed_time(ev1) * 1e3
    tot = (buf[4] + buf[5] + buf[6]) / waves
    print(f"{label:26s} {us:7.1f} us | per wave (100 MHz ticks): prologue {buf[4]/waves:6.0f} loop {buf[5]/waves:7.0f} "
          f"(wait {buf[0]/waves:5.0f} bar {buf[1]/waves:5.0f} issue {buf[2]/waves:5.0f} comp {buf[3]/waves:5.0f}) epilogue {buf[6]/waves:6.0f}"
          f"  total {tot:6.0f}; wave-slots/SIMD = {waves/1024:.1f} -> {waves/1024/2*tot/100:.1f} us if 2 waves/SIMD")
run(50688, 1152, 384, 0, 0, "S qkv store")
run(50688, 1536, 384, 0, 1, "S fc1 gelu", aux=False)
run(50688, 1536, 384, 0, 1, "S fc1 gelu+pre", aux=True)
run(50688, 384, 384, 0, 2, "S proj residual")
run(50688, 384, 1536, 0, 2, "S fc2 residual")
run(50688, 1536, 384, 1, 4, "S fc2 dgrad dgelu")
run(50688, 384, 1536, 1, 0, "S fc1 dgrad store")
