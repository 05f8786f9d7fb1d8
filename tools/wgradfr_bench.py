#!/usr/bin/env python3
"""The weight gradients of one student block (qkv 1152x384, proj 384x384, fc1 1536x384, fc2 384x1536 over M = 50688 token rows): the four split-K
launches on 128x128 tiles (rounds 1-5) against ONE launch of the full-row weight-gradient kernel (devit_wgrad_grouped), warm (back to back) and
cold (the Infinity Cache flushed before every timed launch), and the grouped launch over split_k.
With a library built by tools/build_variant.sh wgstamp "-DDEVIT_GEMMFR_STAMP" (DEVIT_LIB_PATH=tools/_diag/libdevit_wgstamp.so): the in-kernel timeline."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L

dev = torch.device("cuda"); M = int(os.environ.get("ROWS", 50688)); BF = torch.bfloat16; D, Hd = 384, 1536
def rnd(*s): return torch.randn(*s, device=dev).to(BF)
big = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
dqkv, ln1, g1, ao, dh, ln2, g2, h = rnd(M, 3 * D), rnd(M, D), rnd(M, D), rnd(M, D), rnd(M, Hd), rnd(M, D), rnd(M, D), rnd(M, Hd)
gw = [torch.zeros(sh, dtype=torch.float32, device=dev) for sh in ((3 * D, D), (D, D), (Hd, D), (D, Hd))]
gb = [torch.zeros(n, dtype=torch.float32, device=dev) for n in (3 * D, D, Hd, D)]
jobs = [(g2, h, gw[3], None), (dh, ln2, gw[2], gb[2]), (g1, ao, gw[1], None), (dqkv, ln1, gw[0], gb[0])]
flops = sum(2.0 * M * w.shape[0] * w.shape[1] for _, _, w, _ in jobs)
nbytes = sum(M * (w.shape[0] + w.shape[1]) * 2 for _, _, w, _ in jobs)

def cold(fn, n=5):
    best = 1e9
    for _ in range(n):
        big.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best

def warm(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def old():
    for j in jobs: ops.linear_wgrad(*j, M)

lib = L.load()
stamped = hasattr(lib, "devit_wgrad_debug_buffer")
if not stamped:
    for name, fn in (("4 x split-K 128x128", old), ("grouped, split auto ", lambda: ops.linear_wgrads(jobs, M))):
        w, c = warm(fn), cold(fn)
        print(f"{name}: warm {w:7.1f} us  cold {c:7.1f} us   {flops / c / 1e6:6.0f} TF/s cold, {nbytes / c / 1e3:6.0f} GB/s of operands", flush=True)
    for sk in [int(a) for a in os.environ.get("SPLITS", "4 6 8 10 12 13").split()]:
        fn = lambda: ops.linear_wgrads(jobs, M, split_k=sk)
        w, c = warm(fn), cold(fn)
        print(f"grouped, split {sk:3d} ({19 * sk:3d} workgroups): warm {w:7.1f} us  cold {c:7.1f} us", flush=True)
    for sub, name in (([jobs[0], jobs[1]], "MLP pair (fc2, fc1)"), ([jobs[2], jobs[3]], "attention pair (proj, qkv)"), ([jobs[1]], "fc1 alone"), ([jobs[3]], "qkv alone")):
        fn = lambda: ops.linear_wgrads(sub, M)
        print(f"grouped, {name}: warm {warm(fn):7.1f} us  cold {cold(fn):7.1f} us", flush=True)
else:
    dbg = torch.zeros(256 * 4 * 8, dtype=torch.int64, device=dev)
    lib.devit_wgrad_debug_buffer.argtypes = [ctypes.c_void_p]
    lib.devit_wgrad_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    sk = int(os.environ.get("SPLIT", 0))
    fn = lambda: ops.linear_wgrads(jobs, M, split_k=sk)
    for mode in ("warm", "cold"):
        for _ in range(3): fn()
        if mode == "cold": big.zero_()
        dbg.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        d = dbg.view(256 * 4, 8).cpu().double(); d = d[d[:, 0] > 0]
        med = lambda x: float(x.median())
        nk = d[:, 0]
        print(f"{mode}: kernel by events (stamped build) {e0.elapsed_time(e1) * 1e3:.1f} us; waves {len(d)}, K-steps per workgroup {int(nk.min())}..{int(nk.max())}")
        print(f"  prologue {med(d[:, 1]):7.0f}   K loop {med(d[:, 2]):8.0f}   epilogue {med(d[:, 5]):8.0f} (max {float(d[:, 5].max()):.0f})   entry->exit {med(d[:, 7] - d[:, 6]):8.0f} cycles")
        print(f"  per K-step {med(d[:, 3] / nk):7.0f} (MFMA-bound 3072), of it in the two barriers' waits {med(d[:, 4] / nk):7.0f}")
        print(f"  exit spread over waves {float(d[:, 7].max() - d[:, 7].min()):.0f} cycles; entry spread {float(d[:, 6].max() - d[:, 6].min()):.0f}", flush=True)
