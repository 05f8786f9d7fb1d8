#!/usr/bin/env python3
"""Diagnostic: K-loop segment shares of the GEMM (needs gpurun_out/libdevit_hip_stamps.so)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from devit_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "_diag", "libdevit_hip_stamps.so")
from devit_amd import ops
lib = L.load()
dev = torch.device("cuda")
def run(M, N, K, a_km, b_km, kind, label, m_valid=0):
    a = torch.randn((K, M) if a_km else (M, K), device=dev).to(torch.bfloat16)
    b = (torch.randn((K, N) if b_km else (N, K), device=dev) * 0.02).to(torch.bfloat16)
    out = torch.zeros((M, N), dtype=torch.float32 if kind in (5, 6) else torch.bfloat16, device=dev)
    buf = (C.c_ulonglong * 8)()
    for i in range(3):
        ops.gemm(a, a.stride(0), a_km, b, b.stride(0), b_km, M, N, K, kind=kind, out=out, ldc=N, split_k=(16 if kind == 5 else 1), m_valid=m_valid)
        torch.cuda.synchronize()
        lib.devit_debug_gemm_stamps(buf, 1)
    tot = sum(buf[:4])
    bm = 256 if M % 256 == 0 else 128
    waves = (M // bm) * (N // 128) * (bm // 32) * (16 if kind == 5 else 1)
    nk = K // 64 // (16 if kind == 5 else 1)
    print(f"{label:28s} wait {buf[0]/tot:5.1%} barrier {buf[1]/tot:5.1%} issue {buf[2]/tot:5.1%} compute {buf[3]/tot:5.1%}"
          f"  cycles/K-step/wave {tot/waves/nk:7.0f}  (wait {buf[0]/waves/nk:5.0f} bar {buf[1]/waves/nk:5.0f} issue {buf[2]/waves/nk:5.0f} comp {buf[3]/waves/nk:5.0f})"
          f" | per wave: prologue {buf[4]/waves:6.0f} loop {buf[5]/waves:7.0f} epilogue {buf[6]/waves:6.0f}")
run(50688, 2304, 768, 0, 0, 0, "T qkv NT")
run(50688, 1536, 384, 0, 0, 0, "S fc1 NT (store)")
run(50688, 384, 1536, 0, 1, 0, "S fc1 dgrad")
run(1536, 384, 50688, 1, 1, 5, "S fc1 wgrad sk16")
print("--- small L2-resident shapes (1 WG per CU)")
run(8192, 2048, 512, 0, 0, 0, "L2 NT 8192x2048x512")
run(8192, 2048, 512, 0, 1, 0, "L2 A_row/B_km")
run(8192, 2048, 512, 1, 1, 6, "L2 A_km/B_km")
run(8192, 2048, 4096, 0, 0, 0, "NT 8192x2048x4096")
print("--- T qkv without output stores (m_valid = 1)")
run(50688, 2304, 768, 0, 0, 0, "T qkv NT, no stores", m_valid=1)
run(50688, 2304, 3072, 0, 0, 0, "50688x2304x3072 NT")
run(50688, 2304, 3072, 0, 0, 0, "50688x2304x3072 NT no st", m_valid=1)
