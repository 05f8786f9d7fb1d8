#!/usr/bin/env python3
"""LayerNorm forward / backward timing at cold caches (a 300 MB fill between launches: in the step the inputs come
from HBM, a back-to-back loop would read them from the 256 MB Infinity Cache)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops
dev = torch.device("cuda"); M = 50688
big = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
def cold(fn, reps=8):
    best = 1e9
    for _ in range(reps):
        big.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
for D in (384, 768):
    x = torch.randn(M, D, device=dev); g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    y = torch.empty(M, D, dtype=torch.bfloat16, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    t = cold(lambda: ops.layernorm_fwd(x, M, D, g, b, 1e-6, y_bf16=y, mean=mean, rstd=rstd))
    print(f"ln_fwd D={D}: {t:6.1f} us  {M*D*6/t/1e6:5.2f} TB/s")
