import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda"); BF = torch.bfloat16
M = 50688
COLD = os.environ.get("COLD", "0") == "1"
_big = torch.empty(320 << 20, dtype=torch.uint8, device=dev) if COLD else None
def timeit(fn, reps=10):
    if COLD:      # operands from HBM, as in the step (flush the Infinity Cache before every timed launch)
        fn(); best = 1e9
        for _ in range(6):
            _big.zero_(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e-3)
        return best
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(reps)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e-3
tag = os.environ.get("DEVIT_GEMM_TILE", "auto")
for name, N, K, kind in (("S qkv store", 1152, 384, 0), ("S proj resid", 384, 384, 2), ("S fc1 gelu+pre", 1536, 384, 1), ("S fc2 resid", 384, 1536, 2),
                         ("T qkv store", 2304, 768, 0), ("T proj resid", 768, 768, 2), ("T fc1 gelu", 3072, 768, 1), ("T fc2 resid", 768, 3072, 2)):
    a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * .02).to(BF); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.float32 if kind == 2 else BF, device=dev); res = torch.randn(M, N, device=dev) if kind == 2 else None
    aux = torch.empty(M, N, dtype=BF, device=dev) if (kind == 1 and name[0] == "S") else None
    t = timeit(lambda: ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, bias=bias, res=res, aux=aux))
    print(f"tile={tag:5s} {name:16s} {2.0*M*N*K/t/1e12:7.1f} TF {t*1e6:7.1f} us", flush=True)
# dgrad dgelu
a = torch.randn(M, 384, device=dev).to(BF); w = (torch.randn(384, 1536, device=dev) * .02).to(BF); pre = torch.randn(M, 1536, device=dev).to(BF); out = torch.empty(M, 1536, dtype=BF, device=dev)
t = timeit(lambda: ops.gemm(a, 384, 0, w, 1536, 1, M, 1536, 384, kind=4, out=out, ldc=1536, aux_in=pre))
print(f"tile={tag:5s} {'S fc2 dgrad dgelu':16s} {2.0*M*1536*384/t/1e12:7.1f} TF {t*1e6:7.1f} us", flush=True)
for name, N, K in (("S fc1 dgrad", 384, 1536), ("S qkv dgrad", 384, 1152), ("S proj dgrad", 384, 384)):
    a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(K, N, device=dev) * .02).to(BF); out = torch.empty(M, N, dtype=BF, device=dev)
    t = timeit(lambda: ops.gemm(a, K, 0, w, N, 1, M, N, K, kind=0, out=out, ldc=N))
    print(f"tile={tag:5s} {name:16s} {2.0*M*N*K/t/1e12:7.1f} TF {t*1e6:7.1f} us", flush=True)
