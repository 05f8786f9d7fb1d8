#!/usr/bin/env python3
"""How long does a collective-sized kernel (32 workgroups x 512 threads, 32 KB LDS each; tools/probe_kernel.hip) wait for a
place on the chip while persistent GEMM grids run back to back on another stream, as a function of devit_set_reserved_cus(n)?
And what do n reserved CUs cost the GEMMs themselves?  (VERDICT r02 next-round #5a; one GPU.)

    hipcc --offload-arch=gfx950 -shared -fPIC -O2 -o tools/_diag/libprobe.so tools/probe_kernel.hip
    python tools/reserve_cus_probe.py > profiles/r03_reserve_cus_probe.txt
"""
import ctypes as C, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from devit_amd import _lib as L, ops

probe = C.CDLL(os.path.join(ROOT, "tools", "_diag", "libprobe.so"))
probe.probe_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda")
M = 50688
shapes = {"teacher fc2 (256x256 tiles, 1 workgroup / CU)": (768, 3072, L.EPI_RESIDUAL_F32),
          "student fc2 (128x128 tiles, 2 workgroups / CU)": (384, 1536, L.EPI_RESIDUAL_F32)}
pbuf = torch.zeros(1 << 20, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
print(f"device: {torch.cuda.get_device_name(0)}; probe kernel = 32 workgroups x 512 threads x 32 KB LDS, ~1M-float pass")
for name, (N, K, kind) in shapes.items():
    a = (torch.randn(M, K, device=dev) * 0.1).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev)
    bias = torch.zeros(N, device=dev)

    def gemm():
        ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, bias=bias, res=res, m_valid=M)
    print(f"\n{name}: M = {M}, N = {N}, K = {K}")
    with torch.cuda.stream(sa):            # clocks and caches settle before anything is timed (the first train of a process
        for _ in range(60):                # ran 15 % slower than every later one)
            gemm()
    torch.cuda.synchronize()
    base_alone = None
    for n in (0, 8, 16, 32, 0, 8, 16, 32):
        L.call("devit_set_reserved_cus", n)
        with torch.cuda.stream(sa):
            for _ in range(6):
                gemm()
        torch.cuda.synchronize()
        # GEMM alone
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sa):
            e0.record()
            for _ in range(12):
                gemm()
            e1.record()
        torch.cuda.synchronize()
        alone = e0.elapsed_time(e1) / 12 * 1e3
        # probe kernel alone
        p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sb):
            probe.probe_launch(pbuf.data_ptr(), pbuf.numel(), 32, 512, 32768, sb.cuda_stream)
            p0.record()
            probe.probe_launch(pbuf.data_ptr(), pbuf.numel(), 32, 512, 32768, sb.cuda_stream)
            p1.record()
        torch.cuda.synchronize()
        base = p0.elapsed_time(p1) * 1e3
        # probe launched at different phases of a 12-launch GEMM train
        waits, trains = [], []
        for rep in range(12):
            start = torch.cuda.Event()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(sa):
                start.record()
                g0.record()
                for _ in range(12):
                    gemm()
                g1.record()
            with torch.cuda.stream(sb):
                sb.wait_event(start)
                torch.cuda._sleep(int((0.15 + 0.19 * rep) * 2.0e6))       # ~0.15 ... 2.2 ms into the train (cycles at ~2 GHz)
                p0.record()
                probe.probe_launch(pbuf.data_ptr(), pbuf.numel(), 32, 512, 32768, sb.cuda_stream)
                p1.record()
            torch.cuda.synchronize()
            waits.append(p0.elapsed_time(p1) * 1e3)
            trains.append(g0.elapsed_time(g1) / 12 * 1e3)
        if n == 0:
            base_alone = alone
        print(f"  reserved {n:2d} CUs: GEMM alone {alone:7.1f} us/launch ({alone / base_alone * 100 - 100:+5.1f} %)", end="")
        print(f" | probe alone {base:5.1f} us; during the GEMM train: median {statistics.median(waits):7.1f}, max {max(waits):7.1f} us"
              f" | GEMM beside the probe {statistics.median(trains):7.1f} us/launch")
L.call("devit_set_reserved_cus", 0)
