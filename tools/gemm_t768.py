import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda")
def bench(M, N, K, kind, reps=10):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = (torch.randn(N, K, device=dev) * .02).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev); out = torch.empty(M, N, dtype=torch.float32 if kind == 2 else torch.bfloat16, device=dev)
    fn = lambda: ops.gemm(a, K, 0, b, K, 0, M, N, K, kind=kind, out=out, ldc=N, res=res if kind == 2 else None)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print(f"tile={os.environ.get('DEVIT_GEMM_TILE','auto')} {M}x{N}x{K} kind {kind}: {2.0*M*N*K/t/1e12:7.1f} TF {t*1e6:7.1f} us", flush=True)
for _ in range(2):
    bench(50688, 768, 3072, 2); bench(50688, 768, 768, 2); bench(50688, 1536, 384, 1); bench(50688, 2304, 768, 0); bench(50688, 3072, 768, 1)
