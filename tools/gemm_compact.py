#!/usr/bin/env python3
"""The GEMM shapes of a physically shrunk student block (shrink_ratio 0.3: 4 of 6 heads -> attention width 256, 1075 -> 1152 of 1536
neurons) next to the dense ones, per tile configuration (DEVIT_GEMM_FORCE=0 auto / 1 128x128 / 3 256x256) and, for the weight
gradients, per split-K factor.  COLD=1: operands from HBM.  One line per shape: TFLOP/s on the executed FLOPs, us."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda"); BF = torch.bfloat16
M = 50688
_big = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
def timeit(fn):
    fn(); best = 1e9
    for _ in range(5):
        _big.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best
tag = "force" + os.environ.get("DEVIT_GEMM_FORCE", "0")
def line(name, flops, t): print(f"{tag} {name:34s} {flops / t / 1e12:7.1f} TF {t * 1e6:7.1f} us", flush=True)
D = 384
for label, Da, Hd in (("dense", 384, 1536), ("compact", 256, 1152)):
    fwd = ((f"{label} qkv  store   N={3*Da} K={D}", 3 * Da, D, 0), (f"{label} proj resid   N={D} K={Da}", D, Da, 2),
           (f"{label} fc1  gelu+pre N={Hd} K={D}", Hd, D, 1), (f"{label} fc2  resid   N={D} K={Hd}", D, Hd, 2))
    for name, N, K, kind in fwd:
        a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * .02).to(BF); bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, dtype=torch.float32 if kind == 2 else BF, device=dev); res = torch.randn(M, N, device=dev) if kind == 2 else None
        aux = torch.empty(M, N, dtype=BF, device=dev) if kind == 1 else None
        line(name, 2.0 * M * N * K, timeit(lambda: ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, bias=bias, res=res, aux=aux)))
    a = torch.randn(M, D, device=dev).to(BF); w = (torch.randn(D, Hd, device=dev) * .02).to(BF); pre = torch.randn(M, Hd, device=dev).to(BF); out = torch.empty(M, Hd, dtype=BF, device=dev)
    line(f"{label} fc2 dgrad dgelu N={Hd} K={D}", 2.0 * M * Hd * D, timeit(lambda: ops.gemm(a, D, 0, w, Hd, 1, M, Hd, D, kind=4, out=out, ldc=Hd, aux_in=pre)))
    for name, N, K in ((f"{label} fc1 dgrad N={D} K={Hd}", D, Hd), (f"{label} qkv dgrad N={D} K={3*Da}", D, 3 * Da), (f"{label} proj dgrad N={Da} K={D}", Da, D)):
        a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(K, N, device=dev) * .02).to(BF); out = torch.empty(M, N, dtype=BF, device=dev)
        line(name, 2.0 * M * N * K, timeit(lambda: ops.gemm(a, K, 0, w, N, 1, M, N, K, kind=0, out=out, ldc=N)))
    if tag == "force0":
        for name, R, Cc in ((f"{label} fc1 wgrad [{Hd},{D}]", Hd, D), (f"{label} fc2 wgrad [{D},{Hd}]", D, Hd), (f"{label} qkv wgrad [{3*Da},{D}]", 3 * Da, D),
                            (f"{label} proj wgrad [{D},{Da}]", D, Da)):
            dy = torch.randn(M, R, device=dev).to(BF); x = torch.randn(M, Cc, device=dev).to(BF); gw = torch.zeros(R, Cc, device=dev); gb = torch.zeros(R, device=dev)
            auto = ops.split_k_for(R, Cc, M // 64)
            for sk in sorted({auto, max(1, auto // 2), auto * 2, max(1, auto * 3 // 4), auto * 3 // 2}):
                t = timeit(lambda: ops.gemm(dy, R, 1, x, Cc, 1, R, Cc, M, kind=L.EPI_ATOMIC_F32, out=gw, ldc=Cc, split_k=sk, aux=gb))
                line(f"{name} split_k={sk}{' (auto)' if sk == auto else ''}", 2.0 * M * R * Cc, t)
