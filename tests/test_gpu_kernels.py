"""Kernel-level parity on a real MI355X: every C-ABI entry point against a plain PyTorch fp32 statement of
the same op on the same (bf16-rounded) inputs.  Tolerances are stated per test: the kernels accumulate in
fp32, so differences are summation-order noise plus one bf16 rounding of the stored output."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from devit_amd import _lib
    _lib.require_device(torch.zeros(1, device="cuda"))
    return torch.device("cuda")


def rnd(shape, dev, std=1.0, seed=0, dtype=F32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * std).to(dev).to(dtype)


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def bf16_ulp_ok(a, ref, extra=0.0):
    """stored-bf16 outputs: within 1 bf16 ulp (2^-8 relative) of the fp32 reference + summation noise."""
    a, ref = a.float(), ref.float()
    tol = ref.abs() * 2 ** -7 + ref.abs().max() * (1e-5 + extra)
    bad = (a - ref).abs() > tol
    assert not bool(bad.any()), f"{int(bad.sum())} elements off; max abs err {float((a - ref).abs().max()):.3e}"


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (384, 256, 192), (1024, 384, 384), (512, 1536, 384)])
def test_gemm_nt_store(dev, M, N, K):
    from devit_amd import ops, _lib as L
    a, w, bias = rnd((M, K), dev, dtype=BF16), rnd((N, K), dev, 0.05, 1, BF16), rnd((N,), dev, 0.1, 2)
    ref = a.float() @ w.float().t() + bias
    out = torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=L.EPI_STORE_BF16, out=out, ldc=N, bias=bias)
    bf16_ulp_ok(out, ref)
    out32 = torch.empty((M, N), dtype=F32, device=dev)
    ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=L.EPI_STORE_F32, out=out32, ldc=N, bias=bias)
    assert relerr(out32, ref) < 2e-5


@pytest.mark.parametrize("M,N,K,kind,mv", [
    (16384, 1152, 384, 0, 0),        # 128x128 tiles: 1152 tiles on <= 512 persistent workgroups (2-3 tiles each, 5+4 n-chunks)
    (12800, 2304, 768, 0, 0),        # 256x256 tiles: 450 tiles on 256 workgroups, uneven shares, K = 12 ring stages per tile
    (4096, 768, 1536, 2, 3900),      # 256x256 residual epilogue with padding rows in the last m-tile
    (9088, 384, 1536, 6, 9000),      # 71 x 3 = 213 tiles (not a multiple of 8), fp32 store, ragged M
    (6400, 1536, 384, 1, 0),         # GELU + pre-activation, many tiles per workgroup
    (8192, 1152, 768, 0, 0),         # 256x256 tiles with a ragged last n-tile (N = 4.5 x 256): clamped B rows, skipped waves
    (2560, 1152, 768, 6, 2500),      # the same, fp32 store, ragged M as well
])
def test_gemm_persistent_schedule(dev, M, N, K, kind, mv):
    """Every output tile of a launch whose workgroups each walk SEVERAL tiles through one continuous LDS ring (cross-tile
    prefetch, tile decode, n-chunked XCD order) against fp32 matmul; rows >= m_valid stay untouched."""
    from devit_amd import ops, _lib as L
    a, w, bias = rnd((M, K), dev, dtype=BF16), rnd((N, K), dev, 0.05, 1, BF16), rnd((N,), dev, 0.1, 2)
    ref = a.float() @ w.float().t() + bias
    lim = mv or M
    if kind == L.EPI_STORE_BF16:
        out = torch.full((M, N), 3.0, dtype=BF16, device=dev)
        ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, bias=bias, m_valid=mv)
        bf16_ulp_ok(out[:lim], ref[:lim])
    elif kind == L.EPI_GELU_BF16:
        out = torch.empty((M, N), dtype=BF16, device=dev)
        pre = torch.empty((M, N), dtype=BF16, device=dev)
        ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, bias=bias, aux=pre)
        bf16_ulp_ok(pre, ref)
        bf16_ulp_ok(out, torch.nn.functional.gelu(ref), extra=1e-4)
    elif kind == L.EPI_RESIDUAL_F32:
        res = rnd((M, N), dev, seed=8)
        out = torch.full((M, N), 3.0, dtype=F32, device=dev)
        ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, bias=bias, res=res, m_valid=mv)
        assert relerr(out[:lim], (res + ref)[:lim]) < 2e-5
    else:
        out = torch.full((M, N), 3.0, dtype=F32, device=dev)
        ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, bias=bias, m_valid=mv)
        assert relerr(out[:lim], ref[:lim]) < 2e-5
    assert bool((out[lim:].float() == 3.0).all())


@pytest.mark.parametrize("M,N,K,kind,mv", [
    (12800, 2304, 768, 0, 0),        # bf16 store: 450 tiles, uneven shares, 12 K-steps per tile
    (8192, 768, 1536, 2, 8000),      # fp32 residual with padding rows in the last m-tile (96 tiles)
    (6400, 1536, 384, 1, 0),         # GELU + pre-activation, 6 K-steps per tile (the shortest loop the step has)
    (16384, 256, 192, 1, 0),         # 3 K-steps: first / middle / last step each run exactly once per tile (64 tiles: the fewest that take 256x256 tiles)
    (16384, 512, 768, 6, 0),         # fp32 store
    (12544, 768, 768, 3, 0),         # patch embedding (64 images x 196 patches): token remap + bias + pos-embed into [B, 198, N]
])
def test_gemm_four_wave_kernel_bit_identical(dev, monkeypatch, M, N, K, kind, mv):
    """The four-wave kernel (gemm4_kernel, K loop = generated inline asm, csrc/gemm4_kloop.inc; the default for plain bf16 stores and the fp32
    residual epilogue at K >= 768, DEVIT_GEMM4=0 / 1 forces it off / on wherever it is built) accumulates every output element in the same order
    as the eight-wave kernel and runs the same epilogue code: outputs bit-identical, twice in a row (the second launch runs on warm caches)."""
    from devit_amd import ops, _lib as L
    a, w, bias = rnd((M, K), dev, dtype=BF16), rnd((N, K), dev, 0.05, 1, BF16), rnd((N,), dev, 0.1, 2)
    res = rnd((M, N), dev, 1.0, 3) if kind == L.EPI_RESIDUAL_F32 else None
    f32 = kind in (L.EPI_RESIDUAL_F32, L.EPI_STORE_F32, L.EPI_PATCH_F32)
    patch = kind == L.EPI_PATCH_F32
    pos = rnd((198, N), dev, 0.02, 4) if patch else None
    outs = {}
    for flag in ("0", "1", "1"):
        monkeypatch.setenv("DEVIT_GEMM4", flag)
        out = torch.full((M // 196 * 198, N) if patch else (M, N), 7.0, dtype=F32 if f32 else BF16, device=dev)
        aux = torch.zeros((M, N), dtype=BF16, device=dev) if kind == L.EPI_GELU_BF16 else None
        ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=kind, out=out, ldc=N, bias=bias, res=res, aux=aux, m_valid=mv, pos=pos,
                 patch_tokens=196 if patch else 0, extra_tokens=2 if patch else 0)
        torch.cuda.synchronize()
        outs.setdefault(flag, []).append((out, aux))
    ref, ref_aux = outs["0"][0]
    if patch:   # the reference itself against torch: rows 2.. of every image = patches @ w^T + bias + pos[2:], rows 0, 1 (the tokens) untouched
        want = (a.float() @ w.float().t() + bias).view(-1, 196, N) + pos[2:]
        assert relerr(ref.view(-1, 198, N)[:, 2:], want) < 2e-5 and bool((ref.view(-1, 198, N)[:, :2] == 7.0).all())
    for out, aux in outs["1"]:
        assert torch.equal(out, ref)
        if aux is not None:
            assert torch.equal(aux, ref_aux)


@pytest.mark.parametrize("M,K,kind,mv", [
    (50688, 1152, 0, 0),      # qkv dgrad: bf16 store, 198 tiles (one per workgroup)
    (12800, 1536, 0, 12700),  # fc1 dgrad, ragged M
    (76800, 384, 0, 0),       # proj dgrad shape, 300 tiles on 256 workgroups: some walk two (A requests across the tile boundary, B cyclic)
    (50688, 384, 2, 50500),   # student proj through a k-major weight: fp32 residual + DropPath scale, 6 K-steps, ragged M
    (12800, 1536, 2, 0),      # student fc2
    (16384, 192, 2, 0),       # 3 K-steps: first / last-but-one / last step, the loop body never runs
    (76800, 256, 2, 0),       # two tiles per workgroup, 4 K-steps
])
def test_gemm_full_row_kernel_bit_identical(dev, monkeypatch, M, K, kind, mv):
    """The full-row 256x384 kernel (gemmfr_kernel: row-major activation x K-MAJOR weight, N = 384; K loop = generated inline asm,
    csrc/gemmfr_kloop.inc) accumulates every output element in the same order as the 128x128 kernels and runs the same epilogue code:
    bit-identical to them (DEVIT_GEMMFR=0: the bf16 store on the same k-major weight, the fp32 residual on its row-major transpose --
    same products, same order), twice in a row; rows >= m_valid stay untouched."""
    from devit_amd import ops, _lib as L
    N = 384
    a, bias = rnd((M, K), dev, dtype=BF16), rnd((N,), dev, 0.1, 2)
    wk = rnd((K, N), dev, 0.05, 1, BF16)                       # k-major: [K][N]
    resid = kind == L.EPI_RESIDUAL_F32
    res = rnd((M, N), dev, 1.0, 3) if resid else None
    drop = (rnd((M // 198 + 1,), dev, seed=5) > -1.0).float() * 1.25 if resid else None
    kw = dict(kind=kind, ldc=N, bias=bias if resid else None, res=res, rowscale=drop, rows_per_scale=198 if resid else 0, m_valid=mv)
    def run(flag):
        monkeypatch.setenv("DEVIT_GEMMFR", flag)
        out = torch.full((M, N), 7.0, dtype=F32 if resid else BF16, device=dev)
        if flag == "0" and resid:
            ops.gemm(a, K, 0, wk.t().contiguous(), K, 0, M, N, K, out=out, **kw)
        else:
            ops.gemm(a, K, 0, wk, N, 1, M, N, K, out=out, **kw)
        torch.cuda.synchronize()
        return out
    ref = run("0")
    want = a.float() @ wk.float()
    lim = mv or M
    if resid:
        want = res + drop.repeat_interleave(198)[:M, None] * (want + bias)
        assert relerr(ref[:lim], want[:lim]) < 2e-5
    else:
        bf16_ulp_ok(ref[:lim], want[:lim])
    for _ in range(2):
        assert torch.equal(run("1"), ref)
    assert bool((ref[lim:].float() == 7.0).all())


@pytest.mark.parametrize("R,Cc", [(384, 1536), (100, 72), (64, 64)])
def test_transpose16(dev, R, Cc):
    """devit_index_copy mode 4: dst[c][r] = src[r][c] on 16-bit elements (the k-major copy of a Linear weight), ragged edges included."""
    from devit_amd import ops
    src = rnd((R, Cc), dev, dtype=BF16)
    dst = torch.zeros((Cc, R), dtype=BF16, device=dev)
    ops.transpose16(src, dst)
    assert torch.equal(dst, src.t())


def test_gemm_ragged_gelu_dgelu(dev):
    """N = 1152 (4.5 x 256: the compacted student's hidden width at shrink_ratio 0.3) through the 256x256 tile with a half-empty
    last n-tile, GELU (+ pre-activation, + gate) and dGELU epilogues."""
    from devit_amd import ops, _lib as L
    M, D, Hd = 5120, 384, 1152
    x, w1, b1 = rnd((M, D), dev, dtype=BF16), rnd((Hd, D), dev, 0.08, 1, BF16), rnd((Hd,), dev, 0.1, 2)
    gate = (rnd((Hd,), dev, seed=4) > -0.5).float()
    pre_ref = x.float() @ w1.float().t() + b1
    h = torch.full((M, Hd), 3.0, dtype=BF16, device=dev)
    pre = torch.full((M, Hd), 3.0, dtype=BF16, device=dev)
    ops.gemm(x, D, 0, w1, D, 0, M, Hd, D, kind=L.EPI_GELU_BF16, out=h, ldc=Hd, bias=b1, colscale=gate, aux=pre, m_valid=5000)
    bf16_ulp_ok(pre[:5000], pre_ref[:5000])
    bf16_ulp_ok(h[:5000], (torch.nn.functional.gelu(pre_ref) * gate)[:5000], extra=1e-4)
    assert bool((h[5000:].float() == 3.0).all()) and bool((pre[5000:].float() == 3.0).all())
    w2 = rnd((D, Hd), dev, 0.05, 6, BF16)
    dyv = rnd((M, D), dev, seed=9, dtype=BF16)
    dh = torch.full((M, Hd), 3.0, dtype=BF16, device=dev)
    ops.gemm(dyv, D, 0, w2, Hd, 1, M, Hd, D, kind=L.EPI_DGELU_BF16, out=dh, ldc=Hd, colscale=gate, aux_in=pre, m_valid=5000)
    p32 = pre.float().requires_grad_(True)
    torch.nn.functional.gelu(p32).sum().backward()
    bf16_ulp_ok(dh[:5000], ((dyv.float() @ w2.float()) * gate * p32.grad)[:5000], extra=1e-4)
    assert bool((dh[5000:].float() == 3.0).all())


def test_gemm_wgrad_persistent(dev):
    """Split-K atomic epilogue with more (tile, slice) work items than persistent workgroups: the ring restarts on every
    tile (the epilogue stages through its LDS)."""
    from devit_amd import ops, _lib as L
    rows, N, K, split = 16384, 1536, 384, 20                       # 12 x 3 tiles x 20 slices = 720 work items
    dy, x = rnd((rows, N), dev, dtype=BF16), rnd((rows, K), dev, seed=3, dtype=BF16)
    out = torch.zeros((N, K), dtype=F32, device=dev)
    ops.gemm(dy, N, 1, x, K, 1, N, K, rows, kind=L.EPI_ATOMIC_F32, out=out, ldc=K, split_k=split)
    assert relerr(out, dy.float().t() @ x.float()) < 2e-5


@pytest.mark.parametrize("rows,N,K,split", [(16384, 1536, 384, 20), (50688, 1152, 384, 18), (2560, 384, 1536, 3),
                                            (1024, 128, 128, 1)])
def test_gemm_wgrad_fused_bias_grad(dev, rows, N, K, split):
    """aux of the split-K epilogue: b_grad[n] += sum_m dy[m][n] from the same launch (row sums of the A fragments),
    accumulated onto what the buffer held; integer-valued dy -> the sums are exact in fp32."""
    from devit_amd import ops, _lib as L
    g = torch.Generator(device="cpu").manual_seed(rows + N)
    dy = torch.randint(-3, 4, (rows, N), generator=g).to(BF16).to(dev)
    x = rnd((rows, K), dev, seed=3, dtype=BF16)
    out = torch.zeros((N, K), dtype=F32, device=dev)
    bg = torch.full((N,), 2.0, dtype=F32, device=dev)
    ops.gemm(dy, N, 1, x, K, 1, N, K, rows, kind=L.EPI_ATOMIC_F32, out=out, ldc=K, split_k=split, aux=bg)
    assert relerr(out, dy.float().t() @ x.float()) < 2e-5
    assert torch.equal(bg, dy.float().sum(0) + 2.0)
    # random (non-integer) data against the fp32 column sums
    dy = rnd((rows, N), dev, seed=5, dtype=BF16)
    bg.zero_()
    out.zero_()
    ops.gemm(dy, N, 1, x, K, 1, N, K, rows, kind=L.EPI_ATOMIC_F32, out=out, ldc=K, split_k=split, aux=bg)
    assert relerr(bg, dy.float().sum(0)) < 2e-5


def _int_bf16(shape, seed, dev, lo=-3, hi=4):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(BF16).to(dev)


@pytest.mark.parametrize("rows,split", [(16384, 0), (4096, 5), (192 * 64, 64)])
def test_wgrad_grouped_exact_integers(dev, rows, split):
    """devit_wgrad_grouped (the full-row weight-gradient kernel, k-major x k-major): the four products of a student block in ONE launch --
    qkv (1152 outputs: 4.5 tiles), proj (384: 1.5 tiles), fc1 (1536) and fc2 TRANSPOSED (tiles over the hidden features) -- with
    integer-valued operands: every product and every fused bias gradient must be EXACT, on top of what the accumulators held."""
    from devit_amd import ops
    D, Hd = 384, 1536
    dqkv, ln1 = _int_bf16((rows, 3 * D), 1, dev), _int_bf16((rows, D), 2, dev)
    g1, ao = _int_bf16((rows, D), 3, dev), _int_bf16((rows, D), 4, dev)
    dh, ln2 = _int_bf16((rows, Hd), 5, dev), _int_bf16((rows, D), 6, dev)
    g2, h = _int_bf16((rows, D), 7, dev), _int_bf16((rows, Hd), 8, dev)
    gw = [torch.full(sh, 3.0, dtype=F32, device=dev) for sh in ((3 * D, D), (D, D), (Hd, D), (D, Hd))]
    gb = [torch.full((n,), -2.0, dtype=F32, device=dev) for n in (3 * D, D, Hd, D)]
    jobs = [(dqkv, ln1, gw[0], gb[0]), (g1, ao, gw[1], None), (dh, ln2, gw[2], gb[2]), (g2, h, gw[3], gb[3])]
    assert ops.wgrad_jobs_ok(rows, jobs)
    ops.linear_wgrads(jobs, rows, split_k=split)
    torch.cuda.synchronize()
    for (dy, x, w, b), name in zip(jobs, ("qkv", "proj", "fc1", "fc2")):
        ref = dy.float().t() @ x.float() + 3.0
        assert torch.equal(w, ref), (name, float((w - ref).abs().max()))
        if b is not None:
            assert torch.equal(b, dy.float().sum(0) - 2.0), name
    assert bool((gb[1] == -2.0).all())


def test_wgrad_grouped_random_and_compact_shapes(dev):
    """Random data against the fp32 product (summation-order noise only), on the compacted student's shapes: hidden 1152, attention width 256
    (qkv 768 outputs; proj's product transposed: tiles over its 256 input features), leading dimensions wider than the columns read."""
    from devit_amd import ops
    rows, D = 8192, 384
    dqkv, ln1 = rnd((rows, 768), dev, seed=1, dtype=BF16), rnd((rows, D), dev, seed=2, dtype=BF16)
    g1, ao = rnd((rows, D), dev, seed=3, dtype=BF16), rnd((rows, 512), dev, seed=4, dtype=BF16)[:, :256]     # ld 512, 256 columns read
    dh, ln2 = rnd((rows, 1152), dev, seed=5, dtype=BF16), rnd((rows, D), dev, seed=6, dtype=BF16)
    g2, h = rnd((rows, D), dev, seed=7, dtype=BF16), rnd((rows, 1152), dev, seed=8, dtype=BF16)
    gw = [torch.zeros(sh, dtype=F32, device=dev) for sh in ((768, D), (D, 256), (1152, D), (D, 1152))]
    gb = [torch.zeros(n, dtype=F32, device=dev) for n in (768, D, 1152, D)]
    jobs = [(dqkv, ln1, gw[0], gb[0]), (g1, ao, gw[1], gb[1]), (dh, ln2, gw[2], gb[2]), (g2, h, gw[3], None)]
    ops.linear_wgrads(jobs, rows)
    for (dy, x, w, b), name in zip(jobs, ("qkv", "proj", "fc1", "fc2")):
        assert relerr(w, dy.float().t() @ x.float()) < 2e-5, name
        if b is not None:
            assert relerr(b, dy.float().sum(0)) < 2e-5, name


def test_gemm_layout_asymmetric(dev):
    """A = I-like selector with an asymmetric B catches swapped row/col maps (cdna guide §3)."""
    from devit_amd import ops, _lib as L
    M = N = 128
    K = 128
    a = torch.zeros((M, K), dtype=BF16, device=dev)
    a[torch.arange(M), torch.arange(M) % K] = 1
    w = (torch.arange(N, device=dev)[:, None] * 3 + torch.arange(K, device=dev)[None, :] % 7).to(BF16)
    out = torch.empty((M, N), dtype=F32, device=dev)
    ops.gemm(a, K, 0, w, K, 0, M, N, K, kind=L.EPI_STORE_F32, out=out, ldc=N)
    assert torch.equal(out, a.float() @ w.float().t())


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (640, 384, 1536), (384, 1536, 384)])
def test_gemm_dgrad_kmajor_b(dev, M, N, K):
    """out[M, N] = dy[M, K] @ W[K, N]  with W read k-major (transposed LDS reads)."""
    from devit_amd import ops, _lib as L
    dy, w = rnd((M, K), dev, dtype=BF16), rnd((K, N), dev, 0.05, 1, BF16)
    out = torch.empty((M, N), dtype=F32, device=dev)
    ops.gemm(dy, K, 0, w, N, 1, M, N, K, kind=L.EPI_STORE_F32, out=out, ldc=N)
    assert relerr(out, dy.float() @ w.float()) < 2e-5


def test_gemm_kmajor_exact_integers(dev):
    from devit_amd import ops, _lib as L
    M, N, K = 128, 256, 192
    a = ((torch.arange(K, device=dev)[:, None] * 5 + torch.arange(M, device=dev)[None, :]) % 11 - 5).to(BF16)  # [K][M]
    b = ((torch.arange(K, device=dev)[:, None] * 3 + torch.arange(N, device=dev)[None, :] * 7) % 13 - 6).to(BF16)  # [K][N]
    out = torch.zeros((M, N), dtype=F32, device=dev)
    ops.gemm(a, M, 1, b, N, 1, M, N, K, kind=L.EPI_ATOMIC_F32, out=out, ldc=N, split_k=3)
    assert torch.equal(out, a.float().t() @ b.float())


@pytest.mark.parametrize("rows,N,K,split", [(1024, 384, 384, 4), (3200, 1152, 384, 7), (1584 + 80, 128, 256, 1)])
def test_gemm_wgrad_splitk_atomic(dev, rows, N, K, split):
    """dW[N, K] += dy[rows, N]^T @ x[rows, K]: both operands k-major, split-K with fp32 atomics."""
    from devit_amd import ops, _lib as L
    rows = rows // 64 * 64
    dy, x = rnd((rows, N), dev, dtype=BF16), rnd((rows, K), dev, seed=3, dtype=BF16)
    base = rnd((N, K), dev, seed=5)
    out = base.clone()
    ops.gemm(dy, N, 1, x, K, 1, N, K, rows, kind=L.EPI_ATOMIC_F32, out=out, ldc=K, split_k=split)
    ref = base + dy.float().t() @ x.float()
    assert relerr(out, ref) < 2e-5


def test_gemm_epilogues(dev):
    from devit_amd import ops, _lib as L
    B, T, D, Hd = 4, 96, 128, 512
    M = B * T
    x, w1, b1 = rnd((M, D), dev, dtype=BF16), rnd((Hd, D), dev, 0.08, 1, BF16), rnd((Hd,), dev, 0.1, 2)
    gate = (rnd((Hd,), dev, seed=4) > -0.5).float()
    pre_ref = x.float() @ w1.float().t() + b1
    h = torch.empty((M, Hd), dtype=BF16, device=dev)
    pre = torch.empty((M, Hd), dtype=BF16, device=dev)
    ops.gemm(x, D, 0, w1, D, 0, M, Hd, D, kind=L.EPI_GELU_BF16, out=h, ldc=Hd, bias=b1, colscale=gate, aux=pre)
    bf16_ulp_ok(pre, pre_ref)
    bf16_ulp_ok(h, torch.nn.functional.gelu(pre_ref) * gate, extra=1e-4)
    # residual + per-sample rowscale
    w2, b2 = rnd((D, Hd), dev, 0.05, 6, BF16), rnd((D,), dev, 0.1, 7)
    res, rs = rnd((M, D), dev, seed=8), torch.tensor([0.0, 1.25, 1.0, 1.25], device=dev)
    out = torch.empty((M, D), dtype=F32, device=dev)
    att = torch.empty((M, D), dtype=BF16, device=dev)
    ops.gemm(h, Hd, 0, w2, Hd, 0, M, D, Hd, kind=L.EPI_RESIDUAL_F32, out=out, ldc=D, bias=b2, res=res, rowscale=rs,
             rows_per_scale=T, aux=att)
    br = h.float() @ w2.float().t() + b2
    assert relerr(out, res + rs.repeat_interleave(T)[:, None] * br) < 2e-5
    bf16_ulp_ok(att, br)
    # dgelu: out = acc * gate * gelu'(pre)
    dyv = rnd((M, D), dev, seed=9, dtype=BF16)
    dh = torch.empty((M, Hd), dtype=BF16, device=dev)
    ops.gemm(dyv, D, 0, w2, Hd, 1, M, Hd, D, kind=L.EPI_DGELU_BF16, out=dh, ldc=Hd, colscale=gate, aux_in=pre)
    p32 = pre.float().requires_grad_(True)
    torch.nn.functional.gelu(p32).sum().backward()
    bf16_ulp_ok(dh, (dyv.float() @ w2.float()) * gate * p32.grad, extra=1e-4)
    # m_valid guard: rows >= m_valid untouched
    out2 = torch.full((M, D), 7.0, dtype=F32, device=dev)
    ops.gemm(h, Hd, 0, w2, Hd, 0, M, D, Hd, kind=L.EPI_STORE_F32, out=out2, ldc=D, m_valid=300)
    assert bool((out2[300:] == 7.0).all()) and relerr(out2[:300], (h.float() @ w2.float().t())[:300]) < 2e-5


def test_gemm_batched_and_patch(dev):
    from devit_amd import ops, _lib as L
    B, N, D = 3, 198, 128
    f = torch.zeros((B * N + 256, 3 * D), dtype=BF16, device=dev)
    f[: B * N] = rnd((B * N, 3 * D), dev, 0.3, dtype=BF16)
    g = torch.empty((B, 256, 256), dtype=F32, device=dev)
    fj = f[:, D:]
    ops.gemm(fj, 3 * D, 0, fj, 3 * D, 0, 256, 256, D, kind=L.EPI_STORE_F32, out=g, ldc=256, batch=B, a_bs=N * 3 * D,
             b_bs=N * 3 * D, out_bs=65536)
    for b in range(B):
        fb = f[b * N: b * N + N, D:2 * D].float()
        assert relerr(g[b, :N, :N], fb @ fb.t()) < 2e-5
    # patch epilogue: row (b,t) -> token row b*198 + 2 + t, + bias + pos
    Bp, Dm = 2, 128
    rows = torch.zeros((512, 768), dtype=BF16, device=dev)
    rows[: Bp * 196] = rnd((Bp * 196, 768), dev, dtype=BF16)
    w, bias, pos = rnd((Dm, 768), dev, 0.03, 1, BF16), rnd((Dm,), dev, 0.1, 2), rnd((198, Dm), dev, 0.1, 3)
    x = torch.zeros((Bp, 198, Dm), dtype=F32, device=dev)
    ops.gemm(rows, 768, 0, w, 768, 0, 512, Dm, 768, kind=L.EPI_PATCH_F32, out=x, ldc=Dm, bias=bias, pos=pos,
             patch_tokens=196, extra_tokens=2, m_valid=Bp * 196)
    ref = (rows[: Bp * 196].float() @ w.float().t() + bias).view(Bp, 196, Dm) + pos[2:]
    assert relerr(x[:, 2:], ref) < 2e-5 and bool((x[:, :2] == 0).all())


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("D", [384, 768, 192])
def test_layernorm(dev, D):
    from devit_amd import ops
    M = 1000
    x = rnd((M, D), dev, 2.0) + 0.5
    gm, bt = rnd((D,), dev, 0.1, 1) + 1, rnd((D,), dev, 0.1, 2)
    y = torch.empty((M, D), dtype=BF16, device=dev)
    y32 = torch.empty((M, D), dtype=F32, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd(x, M, D, gm, bt, 1e-6, y_bf16=y, y_f32=y32, mean=mean, rstd=rstd)
    xr = x.clone().requires_grad_(True)
    gr, br = gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    assert relerr(y32, ref) < 1e-5
    bf16_ulp_ok(y, ref)
    dy = rnd((M, D), dev, seed=4, dtype=BF16)
    dres = rnd((M, D), dev, seed=5)
    ref.backward(dy.float())
    dx = torch.empty((M, D), dtype=F32, device=dev)
    dxb = torch.empty((M, D), dtype=BF16, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    rsc = torch.tensor([0.5, 2.0, 1.0, 0.0], device=dev)
    gs = torch.ones(D, device=dev)
    ops.layernorm_bwd(dy, False, x, M, D, mean, rstd, gm, dres, dx, dxb, rsc, 250, dg, db, gsum=gs)
    assert relerr(gs, 1 + dxb.float().sum(0)) < 1e-5
    assert relerr(dx, xr.grad + dres) < 1e-5
    bf16_ulp_ok(dxb, (xr.grad + dres) * rsc.repeat_interleave(250)[:, None])
    assert relerr(dg, gr.grad) < 1e-4 and relerr(db, br.grad) < 1e-4
    # row-subset mode (final norm on cls/dist rows only)
    B, T = 5, 198
    xs = rnd((B * T, D), dev, seed=7)
    tok = torch.empty((B * 2, D), dtype=F32, device=dev)
    ops.layernorm_fwd(xs, B * 2, D, gm, bt, 1e-6, y_f32=tok, in_group=2, in_stride=T)
    ref2 = torch.nn.functional.layer_norm(xs.view(B, T, D)[:, :2], (D,), gm, bt, 1e-6)
    assert relerr(tok.view(B, 2, D), ref2) < 1e-5


# ------------------------------------------------------------------------------------------ attention
def attn_ref(qkv, B, N, H, gate):
    D = H * 64
    v = qkv[: B * N].float().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, vv = v[0], v[1], v[2]
    a = ((q @ k.transpose(-2, -1)) * 0.125).softmax(-1)
    o = (a @ vv).transpose(1, 2)
    if gate is not None:
        o = o * gate.view(1, 1, H, 1)
    return o.reshape(B * N, D), a


@pytest.mark.parametrize("B,N,H", [(2, 198, 6), (3, 197, 2), (1, 64, 1)])
def test_attention_fwd_bwd(dev, B, N, H):
    from devit_amd import ops
    from devit_amd._lib import call, ptr, stream_ptr
    D, M = H * 64, B * N
    qkv = ops.rows_alloc(M, 3 * D, BF16, dev, extra=128)
    qkv[:M] = rnd((M, 3 * D), dev, 1.0, dtype=BF16)
    gate = torch.ones(H, device=dev)
    if H > 1:
        gate[1] = 0.0
    out = ops.rows_alloc(M, D, BF16, dev)
    lse = torch.empty((B, H, N), dtype=F32, device=dev)
    call("devit_attn_fwd", ptr(qkv), ptr(out), ptr(lse), ptr(gate), B, N, H, 64, 0.125, 0, stream_ptr())
    q32 = qkv[:M].float().requires_grad_(True)
    ref, _ = attn_ref(q32, B, N, H, gate)
    bf16_ulp_ok(out[:M], ref, extra=2e-3)      # P is rounded to bf16 before P V
    vv = q32.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (vv[0] @ vv[1].transpose(-2, -1)) * 0.125
    assert relerr(lse, torch.logsumexp(s, -1)) < 1e-5
    assert bool((out[M:] == 0).all())
    dout = ops.rows_alloc(M, D, BF16, dev)
    dout[:M] = rnd((M, D), dev, seed=3, dtype=BF16)
    ref.backward(dout[:M].float())
    dqkv = ops.rows_alloc(M, 3 * D, BF16, dev)
    call("devit_attn_bwd", ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(gate), None, ptr(dqkv), B, N, H, 64, 0.125,
         stream_ptr())
    err = relerr(dqkv[:M], q32.grad)
    assert err < 2e-2, f"dqkv rel-to-max err {err:.3e}"    # bf16 P / dS operands: ~2^-8 relative
    # extra gradient added into dqkv (relation loss on the middle block): one rounding, after the fp32 add
    add = ops.rows_alloc(M, 3 * D, BF16, dev)
    add[:M] = rnd((M, 3 * D), dev, 0.5, seed=11, dtype=BF16)
    dqkv2 = ops.rows_alloc(M, 3 * D, BF16, dev)
    call("devit_attn_bwd", ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(gate), ptr(add), ptr(dqkv2), B, N, H, 64, 0.125,
         stream_ptr())
    err2 = relerr(dqkv2[:M], q32.grad + add[:M].float())
    assert err2 < 2e-2, f"dqkv + add rel-to-max err {err2:.3e}"
    bf16_ulp_ok(dqkv2[:M], dqkv[:M].float() + add[:M].float(), extra=2e-3)


# ------------------------------------------------------------------------------------------ elementwise
def test_im2row_embed_cast_colsum(dev):
    from devit_amd import ops
    from devit_amd._lib import call, ptr, stream_ptr
    B = 3
    img = rnd((B, 3, 224, 224), dev)
    rows = ops.rows_alloc(B * 196, 768, BF16, dev)
    call("devit_im2row_bf16", ptr(img), ptr(rows), B, 3, 224, 224, 16, 0, stream_ptr())
    ref = img.reshape(B, 3, 14, 16, 14, 16).permute(0, 2, 4, 1, 3, 5).reshape(B * 196, 768).to(BF16)
    assert torch.equal(rows[: B * 196], ref)
    y = rnd((1000, 384), dev, dtype=BF16)
    out = torch.ones(384, device=dev)
    ops.colsum(y, 1000, 384, out, accumulate=True)
    assert relerr(out, 1 + y.float().sum(0)) < 1e-5
    src = rnd((1001,), dev)
    src16 = torch.empty(1008, dtype=F32, device=dev)[:1001].copy_(src)
    assert torch.equal(ops.cast_bf16(src16), src.to(BF16))


def test_sgemm_small_and_cls_loss(dev):
    from devit_amd import ops
    B, C, D = 16, 25, 384
    tok, w, b = rnd((B, 2, D), dev), rnd((C, D), dev, 0.05, 1), rnd((C,), dev, 0.1, 2)
    lo = torch.empty((B, C), device=dev)
    ops.sgemm_small(tok[:, 1], 2 * D, 1, w, D, 1, b, lo, C, B, C, D)
    assert relerr(lo, tok[:, 1] @ w.t() + b) < 1e-5
    lo_, lk_, lt = rnd((B, C), dev, 1.5, 3), rnd((B, C), dev, 1.5, 4), rnd((B, C), dev, 2.0, 5)
    y = torch.softmax(rnd((B, C), dev, 2.0, 6), -1)
    for kind, tau in (("hard", 1.0), ("soft", 3.0), ("none", 1.0)):
        a, k = lo_.clone().requires_grad_(True), lk_.clone().requires_grad_(True)
        base = torch.sum(-y * torch.log_softmax(a, -1), -1).mean()
        if kind == "hard":
            ref = 0.5 * base + 0.5 * torch.nn.functional.cross_entropy(k, lt.argmax(1))
        elif kind == "soft":
            la, lb = torch.log_softmax(k / tau, 1), torch.log_softmax(lt / tau, 1)
            ref = 0.5 * base + 0.5 * torch.sum(lb.exp() * (lb - la)) * tau * tau / k.numel()
        else:
            ref = base
        ref.backward()
        a2, k2 = lo_.clone().requires_grad_(True), lk_.clone().requires_grad_(True)
        l = ops.ClsDistillLossFn.apply(a2, k2, lt, y, kind, 0.5, tau)
        (l * 2.0).backward()
        assert abs(float(l) - float(ref)) < 1e-5 * max(1, abs(float(ref)))
        assert relerr(a2.grad, 2 * a.grad) < 1e-5
        if kind != "none":
            assert relerr(k2.grad, 2 * k.grad) < 1e-5


def test_relation_loss(dev):
    from devit_amd import ops
    B, N, Hs, Ht = 3, 198, 2, 4
    Ds, Dt = Hs * 64, Ht * 64
    M = B * N
    s_buf = ops.rows_alloc(M, 3 * Ds, BF16, dev, extra=128)
    t_buf = ops.rows_alloc(M, 3 * Dt, BF16, dev, extra=128)
    s_buf[:M] = rnd((M, 3 * Ds), dev, 0.25, 1, BF16)
    t_buf[:M] = rnd((M, 3 * Dt), dev, 0.25, 2, BF16)
    s_buf.requires_grad_(True)
    losses = ops.RelationLossFn.apply(s_buf, t_buf, B, N, 64, 64)
    wts = torch.tensor([0.2, 0.1, 0.3], device=dev) / 12
    (losses * wts).sum().backward()
    s32 = s_buf.detach()[:M].float().requires_grad_(True)
    ref = []
    for j in range(3):
        fs, ft = s32[:, j * Ds:(j + 1) * Ds].view(B, N, Ds), t_buf[:M, j * Dt:(j + 1) * Dt].float().view(B, N, Dt)
        t = torch.log_softmax(ft @ ft.transpose(1, 2) / 8, -1)
        s = torch.log_softmax(fs @ fs.transpose(1, 2) / 8, -1)
        ref.append(torch.sum(t.exp() * (t - s)) / B)
    ref = torch.stack(ref)
    (ref * wts).sum().backward()
    assert relerr(losses, ref) < 1e-4, (losses, ref)
    err = relerr(s_buf.grad[:M], s32.grad)
    assert err < 2e-2, f"relation grad rel-to-max err {err:.3e}"


def test_adamw_and_sumsq(dev):
    from devit_amd._lib import call, ptr, stream_ptr, load
    n = 4096 * 3 + 4
    p, g = rnd((n,), dev), rnd((n,), dev, 0.3, 1)
    m, v, ema = torch.zeros(n, device=dev), torch.zeros(n, device=dev), p.clone()
    p16 = torch.empty(n, dtype=BF16, device=dev)
    ref_p = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    ws = torch.empty(load().devit_sumsq_workspace(), dtype=torch.uint8, device=dev)
    gsq = torch.empty(1, device=dev)
    ema_ref = p.clone()
    for step in (1, 2, 3):
        ref_p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        ema_ref = ema_ref * 0.999 + 0.001 * ref_p.detach()
        call("devit_sumsq_f32", ptr(g), n, ptr(gsq), ptr(ws), ws.numel(), stream_ptr())
        dyn = torch.tensor([1e-3, 1 - 0.9 ** step, 1 - 0.999 ** step], device=dev)
        call("devit_adamw_step", ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), ptr(p16), None, ptr(gsq), ptr(dyn), n, 0.9, 0.999,
             1e-8, 0.05, 1.0, 0.999, 1.0, stream_ptr())
    assert abs(float(gsq) - float((g * g).sum())) < 1e-4 * float((g * g).sum())
    assert relerr(p, ref_p.detach()) < 1e-5 and relerr(ema, ema_ref) < 1e-5
    assert torch.equal(p16, p.to(BF16))


def test_comm_rccl_single_rank(dev):
    """devit_comm_*: the C ABI's RCCL binding (dlopen) on one rank -- unique id, communicator, an in-place SUM all-reduce
    on a side stream (identity at world size 1), destroy.  The multi-rank exchange itself is RCCL's."""
    from devit_amd import ddp
    comm = ddp.RcclComm(rank=0, world=1)
    x = torch.randn(1 << 20, device=dev)
    y = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    comm.all_reduce(y, stream=side)
    comm.all_reduce(y[: 1000], stream=side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    with pytest.raises(Exception):
        comm.all_reduce(y.double())
    comm.destroy()
    comm.destroy()            # idempotent


# ------------------------------------------------------------------------------------------ on-device input stage
def _unrow(rows, B):
    """bf16 patch rows [B*196, 768] -> image layout [B, 3, 224, 224]."""
    return rows[: B * 196].view(B, 14, 14, 3, 16, 16).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, 224, 224)


def test_mixup_cutmix_im2row_and_targets(dev, golden):
    """devit_mix_im2row_bf16 / devit_mix_targets against timm's Mixup(mode='batch') formulas evaluated by
    tests/golden/make_golden.py (mixup.npz; lambda and the box are inputs): the mixed batch, as bf16 patch rows, equals
    bf16(formula) -- bit for bit, one rounding; CutMix is an exact copy inside the box; mode 0 is plain im2row.
    RESTATED, NOT PINNED: timm is absent from the image, so mixup.npz holds SURVEY App. B's restatement of timm's arithmetic,
    not outputs of timm itself (the same holds for the PatchEmbed conv and SoftTargetCrossEntropy stand-ins of make_golden.py)."""
    from oracle.detgen import det_array, det_labels
    from devit_amd import ops
    g = golden("mixup")
    B, C = 4, 10
    img = torch.from_numpy(det_array("mix/img", (B, 3, 224, 224))).to(dev)
    y = torch.from_numpy(det_labels("mix/y", B, C)).to(dev)
    assert np.array_equal(y.cpu().numpy(), g["y"])
    sub = lambda a: a[:, :, ::7, ::5]
    lam, box = float(g["lam"]), [int(v) for v in g["box"]]
    mixed = _unrow(ops.mix_patch_rows(img, 1, lam).rows, B)
    assert torch.equal(sub(mixed).float().cpu(), torch.from_numpy(g["mix_img"]).to(BF16).float())
    cut = _unrow(ops.mix_patch_rows(img, 2, 1.0, box).rows, B)
    assert torch.equal(sub(cut).float().cpu(), torch.from_numpy(g["cut_img"]).to(BF16).float())
    for key, (ys, xs) in (("cut_rows", (slice(28, 32), slice(60, 68))), ("cut_rows2", (slice(139, 143), slice(196, 204)))):
        assert torch.equal(cut[:, :, ys, xs].float().cpu(), torch.from_numpy(g[key]).to(BF16).float())   # the box edges
    full_ref = img.clone()
    full_ref[:, :, box[0]:box[1], box[2]:box[3]] = img.flip(0)[:, :, box[0]:box[1], box[2]:box[3]]
    assert torch.equal(cut, full_ref.to(BF16))                                       # every pixel, not only the samples
    assert torch.equal(mixed, (img * lam + img.flip(0) * (1 - lam)).to(BF16))
    plain = ops.mix_patch_rows(img, 0)
    assert torch.equal(plain.rows, ops.patch_rows(img).rows)
    t_mix = ops.mix_targets(y, C, lam, float(g["smoothing"]))
    t_cut = ops.mix_targets(y, C, float(g["lam_cut"]), float(g["smoothing"]))
    assert relerr(t_mix, torch.from_numpy(g["mix_targets"]).to(dev)) < 1e-6
    assert relerr(t_cut, torch.from_numpy(g["cut_targets"]).to(dev)) < 1e-6
    assert float((t_mix.sum(1) - 1).abs().max()) < 1e-6


def test_models_take_patch_rows(dev):
    """A model fed ops.PatchRows (one im2row pass shared by student, teacher and MultiViT backbones) gives the bits of the
    same model fed the fp32 images."""
    import devit_amd
    from devit_amd import ops
    torch.manual_seed(4)
    m = devit_amd.create_model("dedeit", num_classes=10).to(dev).eval()
    img = rnd((3, 3, 224, 224), dev, seed=9)
    with torch.no_grad():
        a = m(img)
        b = m(ops.patch_rows(img))
    assert torch.equal(a, b)
