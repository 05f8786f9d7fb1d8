#!/usr/bin/env python3
"""Race screen of the attention backward's LDS-DMA ring (counted vmcnt, two barriers per block): full-size launches on fresh
random operands, every launch run twice and bit-compared (the kernel has no atomics: any difference is an ordering bug),
the first also against a float64 reference on a few heads.  usage: attn_race_screen.py [rounds]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops
from devit_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for it in range(rounds):
    B, H = (256, 6) if it % 3 else (128, 12)
    N = 198 if it % 4 else 197
    D = H * 64; M = B * N
    g = torch.Generator(device=dev).manual_seed(1000 + it)
    qkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev, extra=128); qkv[:M] = (torch.randn(M, 3 * D, generator=g, device=dev) * 0.6).to(torch.bfloat16)
    out = ops.rows_alloc(M, D, torch.bfloat16, dev); lse = torch.empty(B, H, N, device=dev)
    call("devit_attn_fwd", ptr(qkv), ptr(out), ptr(lse), None, B, N, H, 64, 0.125, 0, stream_ptr())
    dout = ops.rows_alloc(M, D, torch.bfloat16, dev); dout[:M] = (torch.randn(M, D, generator=g, device=dev) * 0.3).to(torch.bfloat16)
    add = (torch.randn(M + 384, 3 * D, generator=g, device=dev) * 0.01).to(torch.bfloat16) if it % 2 else None
    res = []
    for rep in range(2):
        dqkv = torch.full((ops.pad_rows(M), 3 * D), float("nan"), dtype=torch.bfloat16, device=dev)
        if rep:      # perturb timing: other work in flight on a second stream
            s2 = torch.cuda.Stream()
            with torch.cuda.stream(s2):
                junk = torch.randn(4096, 4096, device=dev).mul_(1.0001)
        call("devit_attn_bwd", ptr(qkv), ptr(out), ptr(dout), ptr(lse), None, ptr(add), ptr(dqkv), B, N, H, 64, 0.125, stream_ptr())
        torch.cuda.synchronize()
        res.append(dqkv[:M].clone())
    same = torch.equal(res[0], res[1])
    finite = bool(torch.isfinite(res[0].float()).all())
    if it == 0:
        b, h = 3, 2
        q, k, v = (qkv[b * N:(b + 1) * N, j * D + h * 64: j * D + (h + 1) * 64].double() for j in range(3))
        do = dout[b * N:(b + 1) * N, h * 64:(h + 1) * 64].double()
        q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
        (torch.softmax(q @ k.t() * 0.125, -1) @ v * do).sum().backward()
        got = res[0][b * N:(b + 1) * N].double()
        err = max(float((got[:, j * D + h * 64: j * D + (h + 1) * 64] - t.grad).abs().max() / t.grad.abs().max()) for j, t in enumerate((q, k, v)))
        print(f"round 0: one head against float64 autograd: rel-to-max {err:.2e}")
        assert err < 2e-2
    bad += (not same) or (not finite)
    print(f"round {it}: B={B} H={H} N={N} add={'y' if add is not None else 'n'} identical={same} finite={finite}", flush=True)
print("attention race screen:", "CLEAN" if bad == 0 else f"{bad} BAD ROUNDS")
sys.exit(1 if bad else 0)
