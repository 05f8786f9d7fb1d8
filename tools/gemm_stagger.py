#!/usr/bin/env python3
"""Does de-phasing the persistent GEMM's workgroups pay?  (Round 2: in-kernel timelines show every workgroup reaching its
epilogue at the same time -- a chip-wide HBM burst with all MFMA pipes idle -- and the store drain exposed in the next tile's
first counted vmcnt wait: 21-36 % of every tile.)  DEVIT_GEMM_STAGGER="cycles,mode" delays workgroup starts by
cycles * phase / 8 (mode 1: only workgroups with a spare tile period, i.e. free; mode 2: all).
Interleaved rounds in ONE process (boxes differ by +-10 %), caches flushed before every timed launch (the state the step
runs in); prints median and min per setting.  usage: python tools/gemm_stagger.py"""
import os, sys, statistics as st
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L

dev = torch.device("cuda")
M, BF = 50688, torch.bfloat16
rnd = lambda *s, dt=BF, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)


def shapes():
    D = 768
    x, xh = rnd(M, D), rnd(M, 4 * D)
    wqkv, w1, w2 = rnd(3 * D, D, std=.02), rnd(4 * D, D, std=.02), rnd(D, 4 * D, std=.02)
    b3, b1, bd = rnd(3 * D, dt=torch.float32), rnd(4 * D, dt=torch.float32), rnd(D, dt=torch.float32)
    o3 = torch.empty(M, 3 * D, dtype=BF, device=dev); o4 = torch.empty(M, 4 * D, dtype=BF, device=dev)
    res32 = rnd(M, D, dt=torch.float32); out32 = torch.empty_like(res32)
    yield "T qkv  store    N=2304 K=768 ", 2.0 * M * 3 * D * D, lambda: ops.gemm(x, D, 0, wqkv, D, 0, M, 3 * D, D, kind=L.EPI_STORE_BF16, out=o3, ldc=3 * D, bias=b3)
    yield "T fc1  gelu     N=3072 K=768 ", 2.0 * M * 4 * D * D, lambda: ops.gemm(x, D, 0, w1, D, 0, M, 4 * D, D, kind=L.EPI_GELU_BF16, out=o4, ldc=4 * D, bias=b1)
    yield "T fc2  residual N=768  K=3072", 2.0 * M * 4 * D * D, lambda: ops.gemm(xh, 4 * D, 0, w2, 4 * D, 0, M, D, 4 * D, kind=L.EPI_RESIDUAL_F32, out=out32, ldc=D, bias=bd, res=res32)
    Ds = 384
    xs = rnd(M, Ds); w1s = rnd(4 * Ds, Ds, std=.02); b1s = rnd(4 * Ds, dt=torch.float32)
    o4s = torch.empty(M, 4 * Ds, dtype=BF, device=dev); o4p = torch.empty_like(o4s)
    yield "S fc1  gelu+pre N=1536 K=384 ", 2.0 * M * 4 * Ds * Ds, lambda: ops.gemm(xs, Ds, 0, w1s, Ds, 0, M, 4 * Ds, Ds, kind=L.EPI_GELU_BF16, out=o4s, ldc=4 * Ds, bias=b1s, aux=o4p)


def once(fn):
    flush.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


SETTINGS = ["0", "20000,1", "40000,1", "80000,1", "160000,1", "10000,2", "20000,2", "40000,2"]
ROUNDS = int(os.environ.get("ROUNDS", "7"))
for name, flops, fn in shapes():
    os.environ.pop("DEVIT_GEMM_STAGGER", None)
    fn(); fn()
    t = {s: [] for s in SETTINGS}
    for _ in range(ROUNDS):
        for s in SETTINGS:                       # interleaved: every setting sees the same box state
            os.environ["DEVIT_GEMM_STAGGER"] = s
            t[s].append(once(fn))
    os.environ.pop("DEVIT_GEMM_STAGGER", None)
    base = st.median(t["0"])
    print(name)
    for s in SETTINGS:
        med, mn = st.median(t[s]), min(t[s])
        print(f"   stagger {s:>9s}: median {med:7.1f} us ({flops / med / 1e6:6.0f} TF, {100 * (base / med - 1):+5.1f} %)   min {mn:7.1f} us", flush=True)
