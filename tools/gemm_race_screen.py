#!/usr/bin/env python3
"""Race screen for the persistent GEMM ring: every (shape, epilogue) is run REPS times on fresh random operands and
compared bit-for-bit with a second run of the same launch and with an fp32 reference; the kernel is deterministic, so
any difference between two runs is an LDS-DMA / ds_read ordering bug that a single passing check can miss."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops, _lib as L
dev = torch.device("cuda"); BF = torch.bfloat16
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 25
cases = [(50688, 1152, 384, 0, 0), (50688, 2304, 768, 0, 0), (50688, 768, 3072, 2, 0), (50688, 1536, 384, 1, 0),
         (50688, 384, 1536, 0, 1), (50688, 1536, 384, 4, 1), (12800, 2304, 768, 0, 0), (9088, 384, 1536, 6, 0),
         # the full-row 256x384 kernel: fc2's forward through a k-major weight (fp32 residual), qkv's / proj's dgrads, two tiles per workgroup
         (50688, 384, 1536, 2, 1), (50688, 384, 1152, 0, 1), (76800, 384, 384, 0, 1)]
bad = 0
for M, N, K, kind, bkm in cases:
    worst = 0.0
    for rep in range(REPS):
        g = torch.Generator(device=dev).manual_seed(rep * 7919 + M + N)
        a = torch.randn(M, K, generator=g, device=dev).to(BF)
        w = (torch.randn((K, N) if bkm else (N, K), generator=g, device=dev) * 0.05).to(BF)
        bias = torch.randn(N, generator=g, device=dev)
        f32 = kind in (2, 6)
        kw = {}
        if kind == 2: kw["res"] = torch.randn(M, N, generator=g, device=dev)
        if kind == 4: kw["aux_in"] = torch.randn(M, N, generator=g, device=dev).to(BF)
        outs = []
        for _ in range(2):
            out = torch.empty(M, N, dtype=torch.float32 if f32 else BF, device=dev)
            ops.gemm(a, K, 0, w, N if bkm else K, bkm, M, N, K, kind=kind, out=out, ldc=N, bias=None if kind == 4 else bias, **kw)
            outs.append(out)
        if not torch.equal(outs[0], outs[1]):
            bad += 1
            print(f"NON-DETERMINISTIC: {M}x{N}x{K} kind {kind} rep {rep}: {int((outs[0] != outs[1]).sum())} elements differ")
        if kind in (0, 6) :
            ref = a.float() @ (w.float() if bkm else w.float().t()) + bias
            worst = max(worst, float((outs[0].float() - ref).abs().max() / ref.abs().max()))
    print(f"{M}x{N}x{K} kind {kind} b_km {bkm}: {REPS} reps ok, worst rel err vs fp32 {worst:.2e}", flush=True)
# the grouped weight-gradient kernel (wgradfr_kernel): (a) one block's four products at M = 50688, 13 K slices: integer-valued operands make every
# product and column sum EXACT whatever the order of the atomics -- any deviation is a ring / staging bug; (b) the products of 11 blocks in one launch,
# no K split: every output element is added once, onto zeros -- two runs on random data must agree bit for bit, and with the fp32 reference to 2e-5
def ints(*s, seed):
    gg = torch.Generator(device=dev).manual_seed(seed)
    return torch.randint(-3, 4, s, generator=gg, device=dev).to(BF)
M, D, Hd = 50688, 384, 1536
shapes = ((3 * D, D), (D, D), (Hd, D), (D, Hd))
for rep in range(REPS):
    ops_ = [(ints(M, n, seed=rep * 31 + i), ints(M, k, seed=rep * 37 + i + 100)) for i, (n, k) in enumerate(shapes)]
    gw = [torch.zeros(sh, dtype=torch.float32, device=dev) for sh in shapes]
    gb = [torch.zeros(sh[0], dtype=torch.float32, device=dev) for sh in shapes]
    ops.linear_wgrads([(dy, x, w, b if i != 3 else None) for i, ((dy, x), w, b) in enumerate(zip(ops_, gw, gb))], M)
    for i, ((dy, x), w, b) in enumerate(zip(ops_, gw, gb)):
        if not torch.equal(w, dy.float().t() @ x.float()) or (i != 3 and not torch.equal(b, dy.float().sum(0))):
            bad += 1
            print(f"WGRAD MISMATCH rep {rep} product {i}: {int((w != dy.float().t() @ x.float()).sum())} elements")
print(f"grouped weight gradients, one block, 13 slices, integer operands: {REPS} reps exact", flush=True)
NB = 11
acts = [[(torch.randn(M, n, device=dev).to(BF), torch.randn(M, k, device=dev).to(BF)) for (n, k) in shapes] for _ in range(2)]     # two blocks' operands, reused
for rep in range(max(2, REPS // 5)):
    runs = []
    for _ in range(2):
        gw = [[torch.zeros(sh, dtype=torch.float32, device=dev) for sh in shapes] for _ in range(NB)]
        jobs = [(acts[l & 1][i][0], acts[l & 1][i][1], gw[l][i], None) for l in range(NB) for i in range(4)]
        ops.linear_wgrads(jobs, M)
        runs.append(gw)
    same = all(torch.equal(a_, b_) for la, lb in zip(*runs) for a_, b_ in zip(la, lb))
    ref_ok = all(float((runs[0][l][i] - acts[l & 1][i][0].float().t() @ acts[l & 1][i][1].float()).abs().max()) <= 2e-5 * float((acts[l & 1][i][0].float().t() @ acts[l & 1][i][1].float()).abs().max())
                 for l in (0, 1, NB - 1) for i in range(4))
    if not same or not ref_ok:
        bad += 1
        print(f"WGRAD 11-block launch rep {rep}: deterministic {same}, reference {ref_ok}")
print(f"grouped weight gradients, {NB} blocks in one launch (no K split): bit-identical across runs, fp32 reference within 2e-5", flush=True)
print("race screen:", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
