"""Tensor-level wrappers over the C ABI (include/devit_hip.h) and the autograd Functions built on them.

PyTorch is plumbing here: it owns device memory and streams, autograd stitches the hand-written
forward/backward kernel sequences together.  Every GEMM, attention, LayerNorm, loss and optimizer kernel is behind the
C ABI; what torch itself still launches per step is glue -- a few dozen small elementwise kernels (zero-fills of fresh
buffers, combining the loss terms, the gradient adds autograd makes where two paths meet, scalar scales, the DropPath
mask draw: rand / floor / div) and ~25 runtime copies, about 0.3 ms of the 29.7 ms serialized step (measured as 0.33 ms
in profiles/r02_e_bench_serial_kernel_stats.csv, before the loss total moved onto the relation-loss vector).

Layout conventions
  * token rows: M = B * N; every bf16 activation that feeds a GEMM lives in a buffer whose row count is
    padded to a multiple of 128 with ZERO pad rows (`rows_alloc`): the GEMM tiles are 128 rows tall and
    the weight-gradient GEMMs reduce over the padded row count.
  * residual stream: fp32 [B, N, D]; branch activations / branch gradients: bf16.
  * weight gradients are ACCUMULATED straight into `param.grad` (fp32, allocated zero on first use), the
    way DDP "main_grad" fusions do; `backward` returns None for parameters and reports finished blocks
    through `grad_ready` callbacks (devit_amd/ddp.py hooks its bucket all-reduce there).
"""
import ctypes as C
import math

import os
import torch

from . import _lib as L
from ._lib import call, ptr, stream_ptr

BF16, F16, F32 = torch.bfloat16, torch.float16, torch.float32
ROW_TILE = 256


def pad_rows(m):
    return (m + ROW_TILE - 1) // ROW_TILE * ROW_TILE


def rows_alloc(m, cols, dtype, device, extra=0):
    """[pad_rows(m) + extra, cols] buffer whose rows >= m are zero."""
    mp = pad_rows(m) + extra
    t = torch.empty((mp, cols), dtype=dtype, device=device)
    if mp > m:
        t[m:].zero_()
    return t


_ws_cache = {}


def workspace(device, nbytes):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 8 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = t
    return t


# ----------------------------------------------------------------------------------------------
# thin wrappers
# ----------------------------------------------------------------------------------------------
def gemm(a, lda, a_km, b, ldb, b_km, M, N, K, *, kind, out, ldc, bias=None, colscale=None, aux=None, aux_in=None,
         res=None, rowscale=None, rows_per_scale=0, pos=None, patch_tokens=0, extra_tokens=0, exact_gelu=0, batch=1,
         a_bs=0, b_bs=0, out_bs=0, m_valid=0, split_k=1, a_group=0, a_skip=0, b_group=0, b_skip=0, dtype16=0):
    if PROFILE is not None:
        return _profiled_gemm(locals())
    A = L.Operand(a.data_ptr(), lda, a_km, a_group, a_skip, a_bs)
    Bo = L.Operand(b.data_ptr(), ldb, b_km, b_group, b_skip, b_bs)
    ep = L.Epilogue(kind, out.data_ptr(), ldc, _p(bias), _p(colscale), _p(aux), _p(aux_in), _p(res), _p(rowscale),
                    rows_per_scale, _p(pos), patch_tokens, extra_tokens, exact_gelu, out_bs, m_valid, dtype16)
    call("devit_gemm_bf16", C.byref(A), C.byref(Bo), M, N, K, batch, split_k, C.byref(ep), stream_ptr())


def _p(t):
    return None if t is None else t.data_ptr()


def full_row_selected(M, N, K, kind=L.EPI_RESIDUAL_F32):
    """devit_gemm_full_row_selected() itself (csrc/gemm.hip; no GPU needed): would (row-major A) x (K-MAJOR B) with N outputs run on the full-row
    256x384 kernel?  (A k-major weight with the fp32 residual epilogue exists on that kernel only.)  Not a restatement: one rule, one parser of
    DEVIT_GEMMFR."""
    return bool(L.load().devit_gemm_full_row_selected(M, N, K, kind))


# Anything that rewrites 16-bit weight copies IN PLACE (behind the parameters' version counters: the fused optimizer kernel, FlatParams.refresh_bf16)
# re-transposes the k-major copies that were derived from them (transpose16_jobs below); copies re-made by de_vit._w16 are re-derived there.
def transpose16(src, dst):
    """dst[c][r] = src[r][c] for 16-bit matrices (devit_index_copy mode 4): the k-major copy of a Linear weight."""
    _Transposes([(src, dst)]).run()


class _Transposes:
    """A device-resident devit_index_job table of 16-bit transposes, one launch."""

    def __init__(self, pairs):
        jobs = [L.IndexJob(s_.data_ptr(), d.data_ptr(), None, s_.shape[0], s_.shape[1], s_.stride(0), d.stride(0), 4, 2) for s_, d in pairs]
        arr = (L.IndexJob * len(jobs))(*jobs)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(pairs[0][0].device)
        self.keep, self.n = pairs, len(jobs)
        self.key = tuple((s_.data_ptr(), d.data_ptr()) for s_, d in pairs)
        self.blocks = max(1, min(144, max((s_.shape[0] + 63) // 64 * ((s_.shape[1] + 63) // 64) for s_, _ in pairs)))   # one 64 x 64 tile per block

    def run(self):
        call("devit_index_copy", ptr(self.table), self.n, self.blocks, stream_ptr())


# bench.py instrumentation: when PROFILE is a list, every GEMM launch is bracketed by events on the current stream
# (the stream the kernel is launched on) and recorded as (template, M, N, K, batch, start_event, end_event).
PROFILE = None


def _profiled_gemm(kw):
    global PROFILE
    rec, PROFILE = PROFILE, None
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gemm(**kw)
        e1.record()
        # (the student's fc2 forward reads a k-major COPY of its weight on the full-row kernel: still a forward Linear layer of the dominant template)
        fwd_kmajor_copy = kw["b_km"] and kw["kind"] == L.EPI_RESIDUAL_F32
        rec.append((("A_km" if kw["a_km"] else "A_row") + "/" + ("B_km" if kw["b_km"] and not fwd_kmajor_copy else "B_row"), kw["M"], kw["N"],
                    kw["K"], kw["batch"], e0, e1))
    finally:
        PROFILE = rec


# Second instrument of bench.py: the bandwidth-bound kernels.  When PROFILE_HBM is a list, each LayerNorm / attention launch
# is bracketed by events on its stream and recorded as (name, algorithmic bytes, start_event, end_event).
PROFILE_HBM = None


def _bracketed(name, nbytes, fn):
    rec = PROFILE_HBM
    if rec is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    rec.append((name, nbytes, e0, e1))


def split_k_for(out_rows, out_cols, ksteps):
    """Split-K factor of a weight-gradient GEMM: exactly one round of resident workgroups (128x128 tiles, two per
    CU).  More slices only add fp32-atomic traffic (64 KB per workgroup at ~1.3 TB/s chip-wide): measured 640 vs 521
    TFLOP/s (fc1, 14 vs 28 slices), 590 vs 460 (qkv, 18 vs 37)."""
    tiles = (out_rows // 128) * (out_cols // 128)     # wgrad always runs 128x128 tiles, two workgroups per CU
    slots = 512
    return max(1, min(ksteps, slots // tiles))


def linear_fwd(x, w, bias, M, *, out, kind=L.EPI_STORE_BF16, **kw):
    """out[M, N] = x[Mp, K] @ w[N, K]^T (+ epilogue).  x is a padded bf16 buffer."""
    N, K = w.shape
    gemm(x, x.stride(0), 0, w, K, 0, pad_rows(M), N, K, kind=kind, out=out, ldc=out.stride(0), bias=bias, m_valid=M,
         **kw)


def linear_dgrad(dy, w, M, *, out, kind=L.EPI_STORE_BF16, **kw):
    """out[M, K] = dy[Mp, N] @ w[N, K]   (w read k-major)."""
    N, K = w.shape
    gemm(dy, dy.stride(0), 0, w, K, 1, pad_rows(M), K, N, kind=kind, out=out, ldc=out.stride(0), m_valid=M, **kw)


def linear_wgrad(dy, x, w_grad, b_grad, M, **kw):
    """w_grad[N, K] += dy[Mp, N]^T @ x[Mp, K];  b_grad[N] += colsum(dy), one launch.  Pad rows of dy are zero."""
    N, K = w_grad.shape
    mp = pad_rows(M)
    # b_grad: the row sums of dy^T come out of the same launch (fragments the MFMAs read anyway; no second pass over dy)
    gemm(dy, dy.stride(0), 1, x, x.stride(0), 1, N, K, mp, kind=L.EPI_ATOMIC_F32, out=w_grad, ldc=K,
         split_k=split_k_for(N, K, mp // 64), aux=b_grad, **kw)


def wgrad_jobs_ok(mp, jobs):
    """Can these weight gradients run as ONE launch of the full-row weight-gradient kernel (devit_wgrad_grouped)?  jobs: (dy, x, w_grad, b_grad)
    as linear_wgrad takes them.  One side of every product must be exactly 384 features wide (the student's D), the other a multiple of 128
    (wgrad_job_struct says which); DEVIT_WGRADFR=0 keeps the split-K launches on 128x128 tiles."""
    return wgrad_enabled(mp) and len(jobs) <= L.WGRAD_MAX_JOBS and all(
        (w.shape[1] == 384 and w.shape[0] % 128 == 0) or (w.shape[0] == 384 and w.shape[1] % 128 == 0) for _, _, w, _ in jobs)


def linear_wgrads(jobs, M, split_k=0):
    """The weight (and bias) gradients of several Linear layers in ONE launch: jobs = [(dy [Mp, N], x [Mp, K], w_grad [N, K], b_grad [N] or None)].
    w_grad += dy^T x; b_grad += colsum(dy).  K == 384: tiles over dy's features; N == 384 (fc2): the product is taken transposed (tiles over x's
    features) and its bias gradient, if asked for, comes from a column-sum pass."""
    mp = pad_rows(M)
    arr = (L.WgradJob * len(jobs))()
    late = []
    for i, (dy, x, w_grad, b_grad) in enumerate(jobs):
        N, K = w_grad.shape
        j = arr[i]
        if K == 384:
            j.a, j.lda, j.a_cols, j.b, j.ldb = dy.data_ptr(), dy.stride(0), N, x.data_ptr(), x.stride(0)
            j.out, j.ldc, j.transposed, j.a_colsum = w_grad.data_ptr(), K, 0, _p(b_grad)
        else:
            j.a, j.lda, j.a_cols, j.b, j.ldb = x.data_ptr(), x.stride(0), K, dy.data_ptr(), dy.stride(0)
            j.out, j.ldc, j.transposed, j.a_colsum = w_grad.data_ptr(), K, 1, None
            if b_grad is not None:
                late.append((dy, N, b_grad))
    call("devit_wgrad_grouped", arr, len(jobs), mp, split_k, stream_ptr())
    for dy, N, b_grad in late:
        colsum(dy, mp, N, b_grad, True)


# bench.py's third instrument: when PROFILE_WGRAD is a list, each grouped weight-gradient launch (DeferredWgrads.flush) is bracketed by events on
# its stream and recorded as (algorithmic flops, algorithmic bytes, start_event, end_event)
PROFILE_WGRAD = None


def wgrad_enabled(mp):
    """DEVIT_WGRADFR=0 keeps every weight gradient on the split-K 128x128 launches (read per call, as devit_block_bwd reads it)."""
    e = os.environ.get("DEVIT_WGRADFR", "")
    return (e == "" or int(e) != 0) and mp % 64 == 0 and mp // 64 >= 3


def colsum(y, M, N, out, accumulate, row_group=0, row_skip=0):
    nbytes = 64 * N * 4
    ws = workspace(y.device, nbytes)
    call("devit_colsum_bf16", ptr(y), M, N, y.stride(0), row_group, row_skip, ptr(out), int(accumulate), ptr(ws),
         ws.numel(), stream_ptr())


def layernorm_fwd(x2d, rows, D, gamma, beta, eps, *, y_bf16=None, y_f32=None, mean=None, rstd=None, in_group=0,
                  in_stride=0, dtype16=0):
    # algorithmic bytes: the fp32 rows in, the normalised rows out (bf16 and / or fp32)
    nbytes = rows * D * (4 + (2 if y_bf16 is not None else 0) + (4 if y_f32 is not None else 0))
    _bracketed("layernorm_fwd", nbytes, lambda: call(
        "devit_layernorm_fwd", ptr(x2d), rows, D, in_group, in_stride, ptr(gamma), ptr(beta), eps, ptr(y_bf16),
        ptr(y_f32), ptr(mean), ptr(rstd), dtype16, stream_ptr()))


def layernorm_bwd(dy, dy_is_f32, x2d, rows, D, mean, rstd, gamma, dres, dx, dx_bf16, rowscale, rows_per_scale, dgamma,
                  dbeta, in_group=0, in_stride=0, gsum=None):
    nbytes = L.load().devit_layernorm_bwd_workspace(rows, D)
    ws = workspace(x2d.device, nbytes)
    # algorithmic bytes: dy, x and the incoming residual gradient in; the fp32 gradient (and its bf16 copy) out
    nbytes = rows * D * ((4 if dy_is_f32 else 2) + 4 + (4 if dres is not None else 0) + (4 if dx is not None else 0) +
                         (2 if dx_bf16 is not None else 0))
    _bracketed("layernorm_bwd", nbytes, lambda: call(
        "devit_layernorm_bwd", ptr(dy), int(dy_is_f32), ptr(x2d), rows, D, in_group, in_stride, ptr(mean), ptr(rstd),
        ptr(gamma), ptr(dres), ptr(dx), ptr(dx_bf16), ptr(rowscale), rows_per_scale, ptr(dgamma), ptr(dbeta),
        ptr(gsum), 1, ptr(ws), ws.numel(), stream_ptr()))


def cast_bf16(src, dst=None, f16=False):
    """fp32 -> bf16 (or IEEE f16: the frozen teacher's GEMM weights) copy of a tensor."""
    src = src.contiguous()
    if dst is None:
        dst = torch.empty(src.shape, dtype=F16 if f16 else BF16, device=src.device)
    call("devit_cast_bf16", ptr(src), ptr(dst), src.numel(), int(dst.dtype == F16), stream_ptr())
    return dst


_ONES1 = {}


def _ones1(dev):
    """A cached fp32 [1.0] on `dev` (the ones-vector operand of the bias-gradient sgemm calls)."""
    t = _ONES1.get(dev)
    if t is None:
        t = _ONES1[dev] = torch.ones(1, dtype=F32, device=dev)
    return t


class GradSlot:
    """Stands where a parameter stands in BlockParams when the block runs on COMPACTED weights (shrink.compact): the
    weight-gradient GEMMs accumulate into `.grad` [compact shape] and BlockParams.finish_grads() adds it into the kept
    rows / columns of the fp32 master's gradient."""
    __slots__ = ("shape", "device", "grad", "requires_grad")

    def __init__(self, shape, device):
        self.shape, self.device, self.grad, self.requires_grad = tuple(shape), device, None, False


def grad_buf(p):
    """fp32 accumulation buffer of a parameter (== param.grad, zero-initialised on first use)."""
    p = getattr(p, "slot", p)             # a compact bias: the value the kernels read + the slot its gradient goes to
    if p.grad is None:
        if isinstance(p, GradSlot):
            p.grad = torch.zeros(p.shape, dtype=F32, device=p.device)
        else:
            p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


def sgemm_small(A, sam, sak, Bm, sbn, sbk, bias, Cm, ldc, M, N, K, alpha=1.0, accumulate=False):
    call("devit_sgemm_small", ptr(A), sam, sak, ptr(Bm), sbn, sbk, ptr(bias), ptr(Cm), ldc, M, N, K, alpha,
         int(accumulate), stream_ptr())


class DeferredWgrads:
    """The weight gradients of SEVERAL blocks in one launch of the full-row weight-gradient kernel (devit_wgrad_grouped).  A block's four
    products are 19 tiles of 256 x 384: alone they need 13 K slices to fill 256 CUs, and the slices' fp32 atomics (97 MB per block at the
    ~1.3 TB/s the memory side adds floats at) are a third of the launch.  Grouped over g blocks the same CUs are filled by 13 / g slices:
    the blocks' backward records its products as jobs (devit_block_bwd_io.defer_jobs / _block_backward(defer=...)) and the group is launched
    -- and its blocks reported to `grad_ready` -- behind the backward of its last block.  Groups (DEVIT_WGRAD_GROUP=auto|all|block):
      auto   with a gradient exchange to overlap (a reducer with world > 1 on `grad_ready`): the blocks of one reducer bucket, so that every
             bucket still leaves as soon as its gradients exist; without one: all blocks of the encoder call (one launch, no K split)
      all    one group        bucket  the reducer's buckets whatever the world size        block  every block alone (round 6's first form)"""

    def __init__(self, cfg, nb):
        self.cfg, self.jobs, self.pending, self.keep = cfg, [], [], []
        hook = cfg.grad_ready
        policy = os.environ.get("DEVIT_WGRAD_GROUP", "auto")
        if policy not in ("auto", "all", "block", "bucket"):
            raise L.DevitError(f"DEVIT_WGRAD_GROUP={policy!r}: auto, all, bucket or block")
        has_buckets = getattr(hook, "bucket_of", None) is not None
        if policy == "auto":
            policy = "bucket" if (has_buckets and getattr(hook, "world", 1) > 1) else "all"
        elif policy == "bucket" and not has_buckets:      # (no reducer on this model: nothing to align with)
            policy = "all"
        self.policy = policy
        self.bucket = [hook.bucket_of(bp.all_params()) for bp in cfg.blocks[:nb]] if policy == "bucket" else None

    def last_of_group(self, i):
        """Is block i the last (lowest) block of its group?  (backward walks i downwards)"""
        if i == 0 or self.policy == "block" or len(self.jobs) + 4 > L.WGRAD_MAX_JOBS:   # (no room for the next block's four)
            return True
        return self.policy == "bucket" and self.bucket[i] != self.bucket[i - 1]

    def add(self, bp, jobs, keep=()):
        """jobs: L.WgradJob structs of block bp, whose backward has just been enqueued; keep: whatever owns the memory they point into"""
        self.jobs += jobs
        self.pending.append(bp)
        self.keep.extend(keep)

    def flush(self, mp):
        if self.jobs:
            arr = (L.WgradJob * len(self.jobs))(*self.jobs)
            rec = PROFILE_WGRAD
            if rec is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            call("devit_wgrad_grouped", arr, len(self.jobs), mp, 0, stream_ptr())
            if rec is not None:
                e1.record()
                cols = sum(j.a_cols for j in self.jobs)
                rec.append((2.0 * mp * cols * 384, mp * (cols + 384 * len(self.jobs)) * 2 + cols * 384 * 4, e0, e1))
        for bp in self.pending:
            bp.finish_grads()
            if self.cfg.grad_ready is not None:
                self.cfg.grad_ready(bp.all_params())
        self.jobs, self.pending, self.keep = [], [], []


def wgrad_job_struct(dy, x, w_grad, b_grad):
    """One product of linear_wgrads() as a devit_wgrad_job (None if the full-row weight-gradient kernel does not take it)."""
    N, K = w_grad.shape
    j = L.WgradJob()
    if K == 384 and N % 128 == 0:
        j.a, j.lda, j.a_cols, j.b, j.ldb = dy.data_ptr(), dy.stride(0), N, x.data_ptr(), x.stride(0)
        j.out, j.ldc, j.transposed, j.a_colsum = w_grad.data_ptr(), K, 0, _p(b_grad)
        return j
    if N == 384 and K % 128 == 0 and b_grad is None:
        j.a, j.lda, j.a_cols, j.b, j.ldb = x.data_ptr(), x.stride(0), K, dy.data_ptr(), dy.stride(0)
        j.out, j.ldc, j.transposed, j.a_colsum = w_grad.data_ptr(), K, 1, None
        return j
    return None


# ----------------------------------------------------------------------------------------------
# encoder blocks (models/de_vit.py:103-121 x depth) as ONE autograd node
# ----------------------------------------------------------------------------------------------
class BlockParams:
    """fp32 parameters + cached bf16 GEMM copies of one Block (see de_vit.Block)."""
    __slots__ = ("n1w", "n1b", "qkv_w", "qkv_b", "proj_w", "proj_b", "n2w", "n2b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                 "qkv_w16", "proj_w16", "fc1_w16", "fc2_w16", "fc2_w16t", "num_heads", "head_gate", "neuron_gate", "dp_prob",
                 "module", "compacted", "masters", "finish")

    def all_params(self):
        """The nn.Parameters behind this block (the masters when the block runs compacted)."""
        if getattr(self, "masters", None) is not None:
            return self.masters
        return [self.n1w, self.n1b, self.qkv_w, self.qkv_b, self.proj_w, self.proj_b, self.n2w, self.n2b, self.fc1_w,
                self.fc1_b, self.fc2_w, self.fc2_b]

    def finish_grads(self):
        """After the block's backward: compacted blocks add their compact weight gradients into the masters'."""
        f = getattr(self, "finish", None)
        if f is not None:
            f()


class EncoderCfg:
    """Non-tensor arguments of EncoderFn."""

    def __init__(self, blocks, training, dp_scales, want_qkv, want_att, want_enc, eps=1e-6, exact_gelu=0,
                 grad_ready=None, qkv_pad_layers=None, lean_tokens=0):
        self.blocks, self.training, self.dp_scales = blocks, training, dp_scales
        self.want_qkv, self.want_att, self.want_enc = want_qkv, want_att, want_enc
        self.eps, self.exact_gelu, self.grad_ready = eps, exact_gelu, grad_ready
        # blocks whose packed qkv output gets the zeroed overhang rows RelationLossFn reads (None: every block)
        self.qkv_pad_layers = qkv_pad_layers
        # > 0: the caller reads only the first `lean_tokens` rows of every image of the encoder output (the class /
        # distillation tokens, models/de_vit.py:286-288) and nothing of the last block besides: that block then runs as
        # _tail_forward / _tail_backward and the output is [B, lean_tokens, D]
        self.lean_tokens = lean_tokens


def _block_forward(x, bp, dp, cfg, need_grad, want_att, pad_qkv=True):
    """x: fp32 [B, N, D] contiguous.  Returns (x_out, saved dict)."""
    B, N, D = x.shape
    M, H, dev = B * N, bp.num_heads, x.device
    x2 = x.view(M, D)
    s = {}
    t16 = 1 if bp.qkv_w16.dtype == F16 else 0       # f16: the frozen teacher's forward (no backward)
    BF16 = F16 if t16 else torch.bfloat16
    if t16 and need_grad:
        raise L.DevitError('precision="f16" is forward-only (frozen teacher): run it under torch.no_grad()')
    ln1 = rows_alloc(M, D, BF16, dev)
    mean1 = torch.empty(M, dtype=F32, device=dev) if need_grad else None
    rstd1 = torch.empty(M, dtype=F32, device=dev) if need_grad else None
    layernorm_fwd(x2, M, D, bp.n1w, bp.n1b, cfg.eps, y_bf16=ln1, mean=mean1, rstd=rstd1, dtype16=t16)
    Da = bp.qkv_w16.shape[0] // 3          # attention width = heads * 64 (== D unless the block was compacted, shrink.py)
    # the attention kernels never read rows >= B*N; the relation-loss windows (RelationLossFn) overhang by up to 128
    qkv = rows_alloc(M, 3 * Da, BF16, dev, extra=128 if pad_qkv else 0)
    linear_fwd(ln1, bp.qkv_w16, bp.qkv_b, M, out=qkv, dtype16=t16)
    attn_o = rows_alloc(M, Da, BF16, dev)
    lse = torch.empty((B, H, N), dtype=F32, device=dev) if need_grad else None
    # algorithmic bytes: q, k, v in, the head outputs out (bf16)
    _bracketed("attention_fwd", M * Da * 2 * 4, lambda: call(
        "devit_attn_fwd", ptr(qkv), ptr(attn_o), ptr(lse), ptr(bp.head_gate), B, N, H, Da // H, (Da // H) ** -0.5,
        t16, stream_ptr()))
    x1 = torch.empty((B, N, D), dtype=F32, device=dev)
    att = torch.empty((M, D), dtype=BF16, device=dev) if want_att else None
    dp1, dp2 = dp if dp is not None else (None, None)
    linear_fwd(attn_o, bp.proj_w16, bp.proj_b, M, out=x1.view(M, D), kind=L.EPI_RESIDUAL_F32, res=x2, rowscale=dp1,
               rows_per_scale=N, aux=att, dtype16=t16)
    ln2 = rows_alloc(M, D, BF16, dev)
    mean2 = torch.empty(M, dtype=F32, device=dev) if need_grad else None
    rstd2 = torch.empty(M, dtype=F32, device=dev) if need_grad else None
    layernorm_fwd(x1.view(M, D), M, D, bp.n2w, bp.n2b, cfg.eps, y_bf16=ln2, mean=mean2, rstd=rstd2, dtype16=t16)
    Hd = bp.fc1_w16.shape[0]
    h = rows_alloc(M, Hd, BF16, dev)
    h_pre = rows_alloc(M, Hd, BF16, dev) if need_grad else None
    linear_fwd(ln2, bp.fc1_w16, bp.fc1_b, M, out=h, kind=L.EPI_GELU_BF16, colscale=bp.neuron_gate, aux=h_pre,
               exact_gelu=cfg.exact_gelu, dtype16=t16)
    x2o = torch.empty((B, N, D), dtype=F32, device=dev)
    w2t = getattr(bp, "fc2_w16t", None)
    if w2t is not None and not t16 and full_row_selected(pad_rows(M), D, Hd):   # k-major weight copy -> the full-row 256x384 GEMM (bit-identical)
        gemm(h, h.stride(0), 0, w2t, w2t.stride(0), 1, pad_rows(M), D, Hd, kind=L.EPI_RESIDUAL_F32, out=x2o.view(M, D), ldc=D, bias=bp.fc2_b,
             m_valid=M, res=x1.view(M, D), rowscale=dp2, rows_per_scale=N)
    else:
        linear_fwd(h, bp.fc2_w16, bp.fc2_b, M, out=x2o.view(M, D), kind=L.EPI_RESIDUAL_F32, res=x1.view(M, D), rowscale=dp2,
                   rows_per_scale=N, dtype16=t16)
    if bp.module is not None and not cfg.lean_tokens:  # shrink contract (core/imp_rank.py:31,108): post-mask values
        bp.module.mlp.neuron_output = h[:M].view(B, N, Hd)
        bp.module.attn.head_output = attn_o[:M].view(B, N, H, Da // H)
    elif bp.module is not None:     # (inside de_vit.lean_tail: not handed out, see _encoder_forward_c)
        bp.module.mlp.neuron_output = None
        bp.module.attn.head_output = None
    if need_grad:
        s = dict(x=x, ln1=ln1, mean1=mean1, rstd1=rstd1, qkv=qkv, attn_o=attn_o, lse=lse, x1=x1, ln2=ln2, mean2=mean2,
                 rstd2=rstd2, h=h, h_pre=h_pre, dp1=dp1, dp2=dp2)
    return x2o, qkv, att, s


def _block_backward(dx, g2, s, bp, cfg, dqkv_add, datt, prev_dp2, want_prev_g, prev_fc2_b=None, g2_bias_done=False, defer=None):
    """dx: fp32 [B,N,D] grad of the block output; g2: bf16 [Mp, D] = bf16(dp2 * dx).
    Returns (dx_in fp32 [B,N,D], g_prev bf16 [Mp,D] = bf16(prev_dp2 * dx_in) or None).
    defer: a DeferredWgrads -- the weight gradients the full-row weight-gradient kernel takes are recorded there as jobs (the caller launches
    them with other blocks' and then calls finish_grads / grad_ready); the others run here on the split-K 128x128 launches."""
    B, N, D = dx.shape
    M, H, dev = B * N, bp.num_heads, dx.device
    Hd = bp.fc1_w.shape[0]
    Da = H * 64                      # attention width (< D when heads were compacted away)
    # ---- MLP branch: x2 = x1 + dp2 * fc2(gate * gelu(fc1(ln2)))
    # Order: the weight gradient that only needs g2 first, then dh_pre's producer and its consumers back to back (dh_pre is
    # 156 MB at B = 256; these GEMMs run 1.2-1.7x slower on operands from cold HBM than from the 256 MB Infinity Cache,
    # tools/gemm_bench.py COLD=1; +0.6 % on the step)
    js, keep = [], []

    def wgrad(dy, x, w_grad, b_grad):
        j = wgrad_job_struct(dy, x, w_grad, b_grad) if (defer is not None and wgrad_enabled(pad_rows(M))) else None
        if j is None:
            linear_wgrad(dy, x, w_grad, b_grad, M)
        else:
            js.append(j)
            keep.extend((dy, x, w_grad, b_grad))
    fc2_bias = None if g2_bias_done else grad_buf(bp.fc2_b)
    if fc2_bias is not None and defer is not None and wgrad_enabled(pad_rows(M)) and D == 384 and Hd % 128 == 0:
        colsum(g2, pad_rows(M), D, fc2_bias, True)      # (the transposed product has no bias-gradient side: devit_block_bwd does the same)
        fc2_bias = None
    wgrad(g2, s["h"], grad_buf(bp.fc2_w), fc2_bias)
    dh_pre = rows_alloc(M, Hd, BF16, dev)
    linear_dgrad(g2, bp.fc2_w16, M, out=dh_pre, kind=L.EPI_DGELU_BF16, colscale=bp.neuron_gate, aux_in=s["h_pre"],
                 exact_gelu=cfg.exact_gelu)
    dln2 = rows_alloc(M, D, BF16, dev)
    linear_dgrad(dh_pre, bp.fc1_w16, M, out=dln2)
    wgrad(dh_pre, s["ln2"], grad_buf(bp.fc1_w), grad_buf(bp.fc1_b))
    dx1 = torch.empty((B, N, D), dtype=F32, device=dev)
    g1 = rows_alloc(M, D, BF16, dev)
    fuse_pb = datt is None        # proj bias gradient = column sums of g1, produced by the same LN-bwd launch
    layernorm_bwd(dln2, False, s["x1"].view(M, D), M, D, s["mean2"], s["rstd2"], bp.n2w, dx.view(M, D), dx1.view(M, D),
                  g1, s["dp1"], N, grad_buf(bp.n2w), grad_buf(bp.n2b), gsum=grad_buf(bp.proj_b) if fuse_pb else None)
    # ---- attention branch: x1 = x + dp1 * proj(gate * attn(qkv(ln1)))
    if datt is not None:  # gradient flowing into the exposed 'attention' output (pre-residual, post-proj)
        g1 = g1 + _pad_like(datt, g1)
    dattn = rows_alloc(M, Da, BF16, dev)
    linear_dgrad(g1, bp.proj_w16, M, out=dattn)
    wgrad(g1, s["attn_o"], grad_buf(bp.proj_w), None if fuse_pb else grad_buf(bp.proj_b))
    dqkv = rows_alloc(M, 3 * Da, BF16, dev)
    # algorithmic bytes: q, k, v, o, do in; dq, dk, dv out (+ the relation-loss gradient that is added in)
    _bracketed("attention_bwd", M * Da * 2 * (8 + (3 if dqkv_add is not None else 0)), lambda: call(
        "devit_attn_bwd", ptr(s["qkv"]), ptr(s["attn_o"]), ptr(dattn), ptr(s["lse"]), ptr(bp.head_gate),
        ptr(dqkv_add), ptr(dqkv), B, N, H, 64, 0.125, stream_ptr()))
    dln1 = rows_alloc(M, D, BF16, dev)
    linear_dgrad(dqkv, bp.qkv_w16, M, out=dln1)
    wgrad(dqkv, s["ln1"], grad_buf(bp.qkv_w), grad_buf(bp.qkv_b))
    if defer is not None:
        defer.add(bp, js, keep)
    dx0 = torch.empty((B, N, D), dtype=F32, device=dev)
    g_prev = rows_alloc(M, D, BF16, dev) if want_prev_g else None
    layernorm_bwd(dln1, False, s["x"].view(M, D), M, D, s["mean1"], s["rstd1"], bp.n1w, dx1.view(M, D), dx0.view(M, D),
                  g_prev, prev_dp2, N, grad_buf(bp.n1w), grad_buf(bp.n1b),
                  gsum=grad_buf(prev_fc2_b) if (g_prev is not None and prev_fc2_b is not None) else None)
    return dx0, g_prev


def _pad_like(t2d, ref):
    out = torch.zeros_like(ref)
    out[: t2d.shape[0]] = t2d.to(ref.dtype)
    return out


def scale_cast(dx, rowscale, N):
    B, _, D = dx.shape
    M = B * N
    g = rows_alloc(M, D, BF16, dx.device)
    call("devit_scale_cast_bf16", ptr(dx), ptr(g), ptr(rowscale), N, M, D, stream_ptr())
    return g


# ----------------------------------------------------------------------------------------------
# The LAST block when only the class / distillation tokens of its output are read (models/de_vit.py:286-288 takes
# x[:, 0], x[:, 1] after the final norm, and engine.py:91-92 takes q/k/v of the middle block only): of the last block
# only K and V are needed on all rows; the Q projection, attention, proj, LN2, fc1 and fc2 run on the B * ntok token
# rows.  Same kernels and per-row arithmetic as _block_forward, so the token rows -- and the logits -- are bit-identical
# to the full block's; the reference computes (and then drops) the other 196 rows per image.
# ----------------------------------------------------------------------------------------------
def _gather_tok(src2d, B, N, ntok, cols, dtype, dev, pad=True):
    """Rows (b, t < ntok) of a [B*N (+pad), cols] matrix as a dense [pad_rows(B*ntok), cols] matrix (zero pad rows)."""
    T = B * ntok
    out = rows_alloc(T, cols, dtype, dev) if pad else torch.empty((T, cols), dtype=dtype, device=dev)
    out[:T].view(B, ntok, cols).copy_(src2d[:B * N].view(B, N, cols)[:, :ntok])
    return out


def _tail_forward(x, bp, dp, cfg, need_grad, ntok):
    """x: fp32 [B, N, D] contiguous (output of the block before the last).  Returns (x2_tok fp32 [B, ntok, D], saved)."""
    B, N, D = x.shape
    M, T, H, dev = B * N, B * ntok, bp.num_heads, x.device
    Da = H * 64                           # attention width (< D when heads were compacted away, shrink.compact)
    qkv_b = getattr(bp.qkv_b, "value", bp.qkv_b)
    t16 = 1 if bp.qkv_w16.dtype == F16 else 0
    dt = F16 if t16 else BF16
    if t16 and need_grad:
        raise L.DevitError('precision="f16" is forward-only (frozen teacher): run it under torch.no_grad()')
    x2 = x.view(M, D)
    ln1 = rows_alloc(M, D, dt, dev)
    mean1 = torch.empty(M, dtype=F32, device=dev) if need_grad else None
    rstd1 = torch.empty(M, dtype=F32, device=dev) if need_grad else None
    layernorm_fwd(x2, M, D, bp.n1w, bp.n1b, cfg.eps, y_bf16=ln1, mean=mean1, rstd=rstd1, dtype16=t16)
    kv = rows_alloc(M, 2 * Da, dt, dev)                                  # (K | V) of every row
    linear_fwd(ln1, bp.qkv_w16[Da:], qkv_b[Da:], M, out=kv, dtype16=t16)
    ln1_tok = _gather_tok(ln1, B, N, ntok, D, dt, dev)
    q_tok = rows_alloc(T, Da, dt, dev)                                   # Q of the token rows
    linear_fwd(ln1_tok, bp.qkv_w16[:Da], qkv_b[:Da], T, out=q_tok, dtype16=t16)
    attn_o = rows_alloc(T, Da, dt, dev)
    lse = torch.empty((B, H, ntok), dtype=F32, device=dev) if need_grad else None
    _bracketed("attention_fwd_rows", (M * 2 * Da + 2 * T * Da) * 2, lambda: call(
        "devit_attn_fwd_rows", ptr(q_tok), Da, ptr(kv), 2 * Da, ptr(attn_o), ptr(lse), ptr(bp.head_gate), B, ntok, N, H,
        64, 0.125, t16, stream_ptr()))
    x_tok = _gather_tok(x2, B, N, ntok, D, F32, dev, pad=False)
    dp1, dp2 = dp if dp is not None else (None, None)
    x1 = torch.empty((B, ntok, D), dtype=F32, device=dev)
    linear_fwd(attn_o, bp.proj_w16, bp.proj_b, T, out=x1.view(T, D), kind=L.EPI_RESIDUAL_F32, res=x_tok, rowscale=dp1,
               rows_per_scale=ntok, dtype16=t16)
    ln2 = rows_alloc(T, D, dt, dev)
    mean2 = torch.empty(T, dtype=F32, device=dev) if need_grad else None
    rstd2 = torch.empty(T, dtype=F32, device=dev) if need_grad else None
    layernorm_fwd(x1.view(T, D), T, D, bp.n2w, bp.n2b, cfg.eps, y_bf16=ln2, mean=mean2, rstd=rstd2, dtype16=t16)
    Hd = bp.fc1_w16.shape[0]
    h = rows_alloc(T, Hd, dt, dev)
    h_pre = rows_alloc(T, Hd, dt, dev) if need_grad else None
    linear_fwd(ln2, bp.fc1_w16, bp.fc1_b, T, out=h, kind=L.EPI_GELU_BF16, colscale=bp.neuron_gate, aux=h_pre,
               exact_gelu=cfg.exact_gelu, dtype16=t16)
    x2o = torch.empty((B, ntok, D), dtype=F32, device=dev)
    linear_fwd(h, bp.fc2_w16, bp.fc2_b, T, out=x2o.view(T, D), kind=L.EPI_RESIDUAL_F32, res=x1.view(T, D), rowscale=dp2,
               rows_per_scale=ntok, dtype16=t16)
    if getattr(bp, "module", None) is not None:
        # the shrink contract's debug views (models/de_vit.py:41,77) of THIS block are not produced on the lean path: drop the ones an
        # earlier full forward left, so a ranking pass run inside lean_tail fails loudly instead of reading stale activations (and the
        # arena they kept alive is released)
        bp.module.mlp.neuron_output = None
        bp.module.attn.head_output = None
    s = None
    if need_grad:
        s = dict(x=x, ln1=ln1, mean1=mean1, rstd1=rstd1, kv=kv, ln1_tok=ln1_tok, q_tok=q_tok, attn_o=attn_o, lse=lse,
                 x1=x1, ln2=ln2, mean2=mean2, rstd2=rstd2, h=h, h_pre=h_pre, dp1=dp1, dp2=dp2, ntok=ntok)
    return x2o, s


def _tail_backward(dx, s, bp, cfg, defer=None):
    """dx: fp32 [B, ntok, D] gradient of _tail_forward's output.  Returns the fp32 [B, N, D] gradient of its input.
    Every product that the full block's backward forms from the 196 untouched rows per image is an exact zero there (their
    output gradient is zero), so the sums here hold the same terms; the token rows of dln1 are produced by one K = 3D GEMM
    like the full path's (one rounding), the other rows by the K = 2D GEMM over (dK | dV) -- the same sum without its zeros."""
    ntok = s["ntok"]
    x = s["x"]
    B, N, D = x.shape
    M, T, H, dev = B * N, B * ntok, bp.num_heads, x.device
    Hd = bp.fc1_w.shape[0]
    Da = H * 64
    dx = dx.contiguous()
    g2 = scale_cast(dx, s["dp2"], ntok)
    # ---- MLP branch on the token rows
    linear_wgrad(g2, s["h"], grad_buf(bp.fc2_w), grad_buf(bp.fc2_b), T)
    dh_pre = rows_alloc(T, Hd, BF16, dev)
    linear_dgrad(g2, bp.fc2_w16, T, out=dh_pre, kind=L.EPI_DGELU_BF16, colscale=bp.neuron_gate, aux_in=s["h_pre"],
                 exact_gelu=cfg.exact_gelu)
    dln2 = rows_alloc(T, D, BF16, dev)
    linear_dgrad(dh_pre, bp.fc1_w16, T, out=dln2)
    linear_wgrad(dh_pre, s["ln2"], grad_buf(bp.fc1_w), grad_buf(bp.fc1_b), T)
    dx1 = torch.empty((B, ntok, D), dtype=F32, device=dev)
    g1 = rows_alloc(T, D, BF16, dev)
    layernorm_bwd(dln2, False, s["x1"].view(T, D), T, D, s["mean2"], s["rstd2"], bp.n2w, dx.view(T, D), dx1.view(T, D),
                  g1, s["dp1"], ntok, grad_buf(bp.n2w), grad_buf(bp.n2b), gsum=grad_buf(bp.proj_b))
    # ---- attention branch: dO on the token rows; dQ there, dK / dV on every row
    dattn = rows_alloc(T, Da, BF16, dev)
    linear_dgrad(g1, bp.proj_w16, T, out=dattn)
    linear_wgrad(g1, s["attn_o"], grad_buf(bp.proj_w), None, T)
    dqkv_tok = rows_alloc(T, 3 * Da, BF16, dev)                          # (dQ | dK | dV) of the token rows
    dkv = rows_alloc(M, 2 * Da, BF16, dev)
    _bracketed("attention_bwd_rows", (M * 4 * Da + 4 * T * Da) * 2, lambda: call(
        "devit_attn_bwd_rows", ptr(s["q_tok"]), Da, ptr(s["kv"]), 2 * Da, ptr(s["attn_o"]), ptr(dattn), ptr(s["lse"]),
        ptr(bp.head_gate), ptr(dqkv_tok), 3 * Da, ptr(dkv), 2 * Da, B, ntok, N, H, 64, 0.125, stream_ptr()))
    dqkv_tok[:T].view(B, ntok, 3 * Da)[:, :, Da:].copy_(dkv[:M].view(B, N, 2 * Da)[:, :ntok])
    dln1 = rows_alloc(M, D, BF16, dev)
    linear_dgrad(dkv, bp.qkv_w16[Da:], M, out=dln1)
    dln1_tok = rows_alloc(T, D, BF16, dev)
    linear_dgrad(dqkv_tok, bp.qkv_w16, T, out=dln1_tok)
    dln1[:M].view(B, N, D)[:, :ntok].copy_(dln1_tok[:T].view(B, ntok, D))
    gw, gb = grad_buf(bp.qkv_w), grad_buf(bp.qkv_b)
    # the one product of this block that reduces over ALL rows (dK | dV against ln1) can ride the encoder's grouped weight-gradient launch
    # (defer: a DeferredWgrads whose single group is launched behind block 0 -- the caller then reports this block with that group)
    j = wgrad_job_struct(dkv, s["ln1"], gw[Da:], gb[Da:]) if (defer is not None and wgrad_enabled(pad_rows(M))) else None
    if j is None:
        linear_wgrad(dkv, s["ln1"], gw[Da:], gb[Da:], M)
    linear_wgrad(dqkv_tok[:, :Da], s["ln1_tok"], gw[:Da], gb[:Da], T)
    if defer is not None:
        defer.add(bp, [j] if j is not None else [], (dkv, s["ln1"], gw, gb))
    dx0 = torch.empty((B, N, D), dtype=F32, device=dev)
    layernorm_bwd(dln1, False, x.view(M, D), M, D, s["mean1"], s["rstd1"], bp.n1w, None, dx0.view(M, D), None, None, 0,
                  grad_buf(bp.n1w), grad_buf(bp.n1b))
    dx0[:, :ntok] += dx1                                                   # the residual path exists on the token rows only
    return dx0


# ----------------------------------------------------------------------------------------------
# composite path: whole blocks per ctypes call (devit_encoder_fwd / devit_block_bwd, csrc/encoder.hip).
# Same kernels, arguments and order as _block_forward / _block_backward above (which stay as the granular path that
# bench.py's per-kernel instrumentation and the rare extra-gradient cases use); one arena per encoder call instead of
# ~15 torch allocations per block.  COMPOSITE = False forces the granular path (A/B tests).
# ----------------------------------------------------------------------------------------------
COMPOSITE = os.environ.get("DEVIT_COMPOSITE", "1") == "1"
# Stream whose allocator pool the encoder arenas come from (None: the current stream).  A forward that runs on a side
# stream but is consumed on the main stream (the frozen teacher, engine._teacher_forward_async) sets this to the
# CONSUMER's stream: a 13 GB arena handed across streams with Tensor.record_stream() comes back to the allocator only when
# the recorded event has completed, i.e. a step later -- the host, which runs ahead, would hipMalloc a fresh arena every
# step (measured: 180 ms of host time per step).  With the arena in the consumer's pool the hand-over is ordered by the
# wait_stream() calls that are there anyway (producer starts after the consumer stream's tail, consumer joins the producer).
ARENA_ALLOC_STREAM = None
_sizes_cache = {}


def _arena(nbytes, dev):
    if ARENA_ALLOC_STREAM is not None and dev.type == "cuda":
        with torch.cuda.stream(ARENA_ALLOC_STREAM):
            return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)
_ACT_DT = {L.ACT_LN1: BF16, L.ACT_QKV: BF16, L.ACT_ATTN_O: BF16, L.ACT_ATT: BF16, L.ACT_LN2: BF16, L.ACT_H: BF16,
           L.ACT_H_PRE: BF16}


def _act_sizes(B, N, D, Da, Hd, flags):
    key = ("a", B, N, D, Da, Hd, flags)
    v = _sizes_cache.get(key)
    if v is None:
        sz = (C.c_size_t * L.ACT_COUNT)()
        call("devit_block_acts_sizes", B, N, D, Da, Hd, flags, sz)
        offs, off = [], 0
        for n in sz:
            offs.append(off)
            off += n
        v = _sizes_cache[key] = (list(sz), offs, off)
    return v


def _bwd_sizes(B, N, D, widths):
    """Transient buffers of devit_block_bwd, sized for the widest of `widths` = {(attn_width, hidden)} (compacted blocks of one
    model differ)."""
    key = ("b", B, N, D, tuple(sorted(widths)))
    v = _sizes_cache.get(key)
    if v is None:
        mx = [0] * L.BWD_COUNT
        for Da, Hd in widths:
            sz = (C.c_size_t * L.BWD_COUNT)()
            call("devit_block_bwd_sizes", B, N, D, Da, Hd, sz)
            mx = [max(a, b) for a, b in zip(mx, sz)]
        offs, off = [], 0
        for n in mx:
            offs.append(off)
            off += n
        v = _sizes_cache[key] = (mx, offs, off)
    return v


def _weights_struct(bp):
    w = L.BlockWeights()
    w.n1w, w.n1b, w.qkv_b, w.proj_b = bp.n1w.data_ptr(), bp.n1b.data_ptr(), bp.qkv_b.data_ptr(), bp.proj_b.data_ptr()
    w.n2w, w.n2b, w.fc1_b, w.fc2_b = bp.n2w.data_ptr(), bp.n2b.data_ptr(), bp.fc1_b.data_ptr(), bp.fc2_b.data_ptr()
    w.qkv_w16, w.proj_w16 = bp.qkv_w16.data_ptr(), bp.proj_w16.data_ptr()
    w.fc1_w16, w.fc2_w16 = bp.fc1_w16.data_ptr(), bp.fc2_w16.data_ptr()
    w.head_gate, w.neuron_gate = _p(bp.head_gate), _p(bp.neuron_gate)
    w.num_heads, w.attn_width, w.hidden = bp.num_heads, bp.qkv_w16.shape[0] // 3, bp.fc1_w16.shape[0]
    w.dtype16 = 1 if bp.qkv_w16.dtype == F16 else 0
    w.fc2_w16t = _p(getattr(bp, "fc2_w16t", None))
    return w


class _EncoderRun:
    """What one composite forward leaves behind for backward: the ctypes argument arrays and the arena they point into."""
    __slots__ = ("weights", "acts", "arena", "x", "dims", "views", "dps", "nb")


def _encoder_forward_composite(x, cfg, need_grad, nb):
    """Blocks [0, nb) of cfg.blocks."""
    B, N, D = x.shape
    M, dev = B * N, x.device
    mp = pad_rows(M)
    weights = (L.BlockWeights * nb)()
    acts = (L.BlockActs * nb)()
    run = _EncoderRun()
    # one allocation per block (uniform sizes: the caching allocator hands the same blocks back every step; a single
    # 8-13 GB arena per encoder call gets split by other requests and re-hipMalloc'ed -- measured 150 ms of host per step)
    run.weights, run.acts, run.arena, run.x, run.dims, run.dps, run.nb = weights, acts, [], x, (B, N, D), cfg.dp_scales, nb
    views = []
    x_ptr = x.data_ptr()
    for i, bp in enumerate(cfg.blocks[:nb]):
        weights[i] = _weights_struct(bp)
        pad = bool(cfg.want_qkv) and (cfg.qkv_pad_layers is None or i in cfg.qkv_pad_layers)
        flags = (L.BLK_SAVE if need_grad else 0) | (L.BLK_QKV_PAD if pad else 0) | (L.BLK_ATT if cfg.want_att else 0)
        Da, Hd = weights[i].attn_width, weights[i].hidden
        BF16 = F16 if weights[i].dtype16 else torch.bfloat16
        if weights[i].dtype16 and need_grad:
            raise L.DevitError('precision="f16" is forward-only (frozen teacher): run it under torch.no_grad()')
        sz, offs, tot = _act_sizes(B, N, D, Da, Hd, flags)
        arena = _arena(tot, dev)
        run.arena.append(arena)
        base = arena.data_ptr()
        a = acts[i]
        a.x = x_ptr
        for j in range(L.ACT_COUNT):
            a.buf[j] = base + offs[j] if sz[j] else None
        dp = cfg.dp_scales[i] if cfg.dp_scales is not None else None
        a.dp1, a.dp2 = (_p(dp[0]), _p(dp[1])) if dp is not None else (None, None)
        a.flags = flags
        x_ptr = a.buf[L.ACT_X2]

        def view(j, rows, cols, dt, arena=arena, offs=offs):
            return arena[offs[j]:offs[j] + rows * cols * dt.itemsize].view(dt).view(rows, cols)
        qkv_rows = mp + (128 if flags & L.BLK_QKV_PAD else 0)
        v = dict(qkv=view(L.ACT_QKV, qkv_rows, 3 * Da, BF16), x2=view(L.ACT_X2, M, D, F32).view(B, N, D),
                 att=view(L.ACT_ATT, M, D, BF16) if cfg.want_att else None)
        v["qkv"]._devit_arena = True          # a view of an arena from ARENA_ALLOC_STREAM's pool (engine._hand_over)
        views.append(v)
        if bp.module is not None and not cfg.lean_tokens:  # shrink contract (core/imp_rank.py:31,108): post-mask values
            bp.module.mlp.neuron_output = view(L.ACT_H, mp, Hd, BF16)[:M].view(B, N, Hd)
            bp.module.attn.head_output = view(L.ACT_ATTN_O, mp, Da, BF16)[:M].view(B, N, bp.num_heads, Da // bp.num_heads)
        elif bp.module is not None:
            # inside de_vit.lean_tail the caller has declared that it reads the logits (and the middle block's q / k / v) only: the
            # debug views are not handed out, and the ones an earlier public forward left are dropped, so a ranking pass run inside
            # lean_tail fails loudly instead of reading stale values.  (This does NOT shrink the 35 GB peak: the per-layer q / k / v
            # views of the returned dict hold the arenas too, and those must stay until the consumer stream has read them -- the
            # arenas of a side-stream forward live in the CONSUMER's pool (ARENA_ALLOC_STREAM); dropping the unread layers' views
            # at once hands their memory to the main stream while the teacher's kernels still write it: measured as a non-finite
            # loss in the two-stream bench, round 4.)
            bp.module.mlp.neuron_output = None
            bp.module.attn.head_output = None
    run.views = views
    call("devit_encoder_fwd", nb, weights, acts, B, N, D, cfg.eps, stream_ptr())
    return run


def _wgrads_struct(bp):
    g = L.BlockWgrads()
    g.n1w, g.n1b, g.qkv_w, g.qkv_b = (grad_buf(p).data_ptr() for p in (bp.n1w, bp.n1b, bp.qkv_w, bp.qkv_b))
    g.proj_w, g.proj_b, g.n2w, g.n2b = (grad_buf(p).data_ptr() for p in (bp.proj_w, bp.proj_b, bp.n2w, bp.n2b))
    g.fc1_w, g.fc1_b, g.fc2_w, g.fc2_b = (grad_buf(p).data_ptr() for p in (bp.fc1_w, bp.fc1_b, bp.fc2_w, bp.fc2_b))
    return g


def _encoder_backward_composite(run, cfg, dx, dqkvs, defer):
    """dx: fp32 [B,N,D] contiguous gradient of the encoder output.  Returns the gradient of the encoder input."""
    B, N, D = run.dims
    M, dev, nb = B * N, dx.device, run.nb
    mp = pad_rows(M)
    sz, offs, tot = _bwd_sizes(B, N, D, {(run.weights[i].attn_width, run.weights[i].hidden) for i in range(nb)})
    # Workspace: the transient buffers shared by all blocks + two fp32 dx buffers that alternate + per-block SLOTS for what a deferred
    # weight-gradient job reads (bf16 g2 = the branch gradient entering the block, dh_pre, g1, dqkv): they must outlive the block's call until
    # its group is launched.  Groups of one block need two slots (block i reads g of slot i, writes the next block's into the other one).
    dxb, gb = M * D * 4, (mp * D * 2 + 255) // 256 * 256
    per = gb + sz[L.BWD_DH_PRE] + sz[L.BWD_G1] + sz[L.BWD_DQKV]
    nslots = 2 if defer.policy == "block" else nb
    ws = torch.empty(tot + 2 * dxb + nslots * per, dtype=torch.uint8, device=dev)
    base = ws.data_ptr()
    dx_ptrs = [base + tot, base + tot + dxb]
    slot0 = tot + 2 * dxb

    def slot(i):                         # byte offsets of (g, dh_pre, g1, dqkv) of block i
        o = slot0 + (i % nslots) * per
        return o, o + gb, o + gb + sz[L.BWD_DH_PRE], o + gb + sz[L.BWD_DH_PRE] + sz[L.BWD_G1]
    g_top = slot(nb - 1)[0]
    call("devit_scale_cast_bf16", ptr(dx), C.c_void_p(base + g_top), ptr(run.dps[nb - 1][1]) if run.dps is not None and run.dps[nb - 1] is not None else None,
         N, M, D, stream_ptr())
    if mp > M:
        ws[g_top + M * D * 2: g_top + mp * D * 2].zero_()
    io = L.BlockBwdIO()
    for j in range(L.BWD_COUNT):
        io.ws[j] = base + offs[j]
    io.lnws_bytes = sz[L.BWD_LNWS]
    got = (L.WgradJob * 4)()
    ngot = C.c_int(0)
    io.defer_jobs, io.defer_count = C.addressof(got), C.addressof(ngot)
    cur_dx, g_bias_done = dx.data_ptr(), 0
    st = stream_ptr()
    for i in range(nb - 1, -1, -1):
        bp = cfg.blocks[i]
        dq = dqkvs[i] if dqkvs else None
        if dq is not None:
            dq = dq.contiguous()
        out_slot = (nb - 1 - i) & 1
        og, oh, o1, oq = slot(i)
        io.dx, io.g2, io.dx_in = cur_dx, base + og, dx_ptrs[out_slot]
        io.ws[L.BWD_DH_PRE], io.ws[L.BWD_G1], io.ws[L.BWD_DQKV] = base + oh, base + o1, base + oq
        prev = cfg.blocks[i - 1] if i > 0 else None
        io.g_prev = base + slot(i - 1)[0] if prev is not None else None
        pdp = run.dps[i - 1] if (prev is not None and run.dps is not None) else None
        io.prev_dp2 = _p(pdp[1]) if pdp is not None else None
        io.prev_fc2_b_grad = grad_buf(prev.fc2_b).data_ptr() if prev is not None else None
        io.g2_bias_done = g_bias_done
        io.dqkv_add = _p(dq)
        wg = _wgrads_struct(bp)
        call("devit_block_bwd", C.byref(run.weights[i]), C.byref(run.acts[i]), C.byref(wg), C.byref(io), B, N, D, cfg.eps, st)
        cur_dx, g_bias_done = dx_ptrs[out_slot], 1 if prev is not None else 0
        defer.add(bp, [L.WgradJob.from_buffer_copy(got[k]) for k in range(ngot.value)], (dq,))
        if defer.last_of_group(i):
            defer.flush(mp)
    o = tot + (0 if cur_dx == dx_ptrs[0] else dxb)
    return ws[o:o + dxb].view(F32).view(B, N, D)


class EncoderFn(torch.autograd.Function):
    """x -> blocks[0..n) -> (x_out, [qkv_i bf16 packed ...], [att_i ...], [enc_i ...])."""

    @staticmethod
    def forward(ctx, x, cfg, *params):
        L.require_device(x)
        x = x.contiguous()
        need_grad = cfg.grad_enabled and (x.requires_grad or any(p is not None and p.requires_grad for p in params))
        if need_grad and any(getattr(bp, "compacted", False) and getattr(bp, "masters", None) is None for bp in cfg.blocks):
            raise L.DevitError("this model was compacted for inference (shrink.compact(model)): run it under torch.no_grad(), "
                               "or compact it with shrink.compact(model, trainable=True) to train through the compacted blocks")
        ctx.run = None
        nb = len(cfg.blocks)
        # lean tail (EncoderCfg.lean_tokens): the last block runs on the token rows only
        lean = cfg.lean_tokens if (cfg.lean_tokens and nb >= 2 and not cfg.want_att and not cfg.want_enc) else 0
        nbody = nb - 1 if lean else nb
        dp_last = cfg.dp_scales[nb - 1] if cfg.dp_scales is not None else None
        ctx.tail = None
        if COMPOSITE and PROFILE is None and PROFILE_HBM is None:
            run = _encoder_forward_composite(x, cfg, need_grad, nbody)
            qkvs = [v["qkv"] for v in run.views] if cfg.want_qkv else []
            atts = [v["att"] for v in run.views] if cfg.want_att else []
            encs = [run.views[i]["x2"].clone() if i == nb - 1 else run.views[i]["x2"] for i in range(nb)] if cfg.want_enc else []
            ctx.cfg, ctx.saved, ctx.need_grad = cfg, None, need_grad
            ctx.run = run if need_grad else None
            ctx.counts = (len(qkvs), len(atts), len(encs))
            ctx.set_materialize_grads(False)
            xo = run.views[-1]["x2"]
            if lean:
                xo, ctx.tail = _tail_forward(xo, cfg.blocks[-1], dp_last, cfg, need_grad, lean)
            return (xo,) + tuple(qkvs) + tuple(atts) + tuple(encs)
        saved, qkvs, atts, encs = [], [], [], []
        for i, bp in enumerate(cfg.blocks[:nbody]):
            dp = cfg.dp_scales[i] if cfg.dp_scales is not None else None
            pad = bool(cfg.want_qkv) and (cfg.qkv_pad_layers is None or i in cfg.qkv_pad_layers)
            x, qkv, att, s = _block_forward(x, bp, dp, cfg, need_grad, cfg.want_att, pad)
            saved.append(s)
            if cfg.want_qkv:
                qkvs.append(qkv)
            if cfg.want_att:
                atts.append(att)
            if cfg.want_enc:
                encs.append(x.clone() if i == len(cfg.blocks) - 1 else x)
        if lean:
            x, ctx.tail = _tail_forward(x, cfg.blocks[-1], dp_last, cfg, need_grad, lean)
        ctx.cfg, ctx.saved, ctx.need_grad = cfg, saved, need_grad
        ctx.counts = (len(qkvs), len(atts), len(encs))
        # outputs nobody differentiates (11 of the 12 qkv tensors in the DEKD step) arrive as None in backward, not as
        # zero tensors: a materialised one is a 117 MB fill plus a 117 MB read in the attention backward, per block
        ctx.set_materialize_grads(False)
        return (x,) + tuple(qkvs) + tuple(atts) + tuple(encs)

    @staticmethod
    def backward(ctx, dx, *dothers):
        cfg, saved = ctx.cfg, ctx.saved
        if not ctx.need_grad:
            raise L.DevitError("EncoderFn.backward called but the forward ran without grad bookkeeping")
        nq, na, ne = ctx.counts
        dqkvs, datts, dencs = dothers[:nq], dothers[nq:nq + na], dothers[nq + na:]
        nb = len(cfg.blocks)
        nparams = 12 * nb
        if ctx.tail is not None:          # lean tail: dx is the [B, ntok, D] gradient of the token rows
            tail, ctx.tail = ctx.tail, None
            last = cfg.blocks[-1]
            if dx is None:
                dx = torch.zeros_like(tail["x1"])
            nb -= 1
            defer = DeferredWgrads(cfg, nb)
            if defer.policy == "all":          # (no exchange to overlap: the last block is reported with the one group)
                dx = _tail_backward(dx, tail, last, cfg, defer)
            else:
                dx = _tail_backward(dx, tail, last, cfg)
                last.finish_grads()
                if cfg.grad_ready is not None:
                    cfg.grad_ready(last.all_params())
        else:
            defer = DeferredWgrads(cfg, nb)
        if ctx.run is not None:
            run = ctx.run
            if any(d is not None for d in datts) or any(d is not None for d in dencs):
                raise L.DevitError("gradients into the exposed 'attention' / 'encoder' outputs run on the granular path only: "
                                   "set devit_amd.ops.COMPOSITE = False for this model call")
            B, N, D = run.dims
            if dx is None:
                dx = torch.zeros((B, N, D), dtype=F32, device=run.x.device)
            dx_in = _encoder_backward_composite(run, cfg, dx.contiguous(), dqkvs if nq else None, defer)
            ctx.run = None
            return (dx_in, None) + (None,) * nparams
        B, N, D = saved[0]["x"].shape
        if dx is None:
            dx = torch.zeros((B, N, D), dtype=F32, device=saved[0]["x"].device)
        dx = dx.contiguous()
        if ne and dencs[nb - 1] is not None:
            dx = dx + dencs[nb - 1]
        g = scale_cast(dx, saved[nb - 1]["dp2"], N)
        g_bias_done = False           # fc2 bias gradient of block i comes fused from block i+1's LN1 backward
        for i in range(nb - 1, -1, -1):
            bp = cfg.blocks[i]
            dq = dqkvs[i] if nq else None
            if dq is not None:
                dq = dq.contiguous()
            da = datts[i] if na else None
            prev_dp2 = saved[i - 1]["dp2"] if i > 0 else None
            extra = dencs[i - 1] if (ne and i > 0 and dencs[i - 1] is not None) else None
            fuse_prev = i > 0 and extra is None
            dx, g = _block_backward(dx, g, saved[i], bp, cfg, dq, da, prev_dp2, want_prev_g=fuse_prev,
                                    prev_fc2_b=cfg.blocks[i - 1].fc2_b if fuse_prev else None, g2_bias_done=g_bias_done, defer=defer)
            g_bias_done = fuse_prev
            if extra is not None:
                dx = dx + extra
                g = scale_cast(dx, prev_dp2, N)
            saved[i] = None
            if defer.last_of_group(i):
                defer.flush(pad_rows(B * N))
        return (dx, None) + (None,) * nparams


# ----------------------------------------------------------------------------------------------
# patch embedding + token assembly (models/de_vit.py:258-264)
# ----------------------------------------------------------------------------------------------
class PatchRows:
    """A batch of images already cut into bf16 patch rows [pad_rows(B*196), 768] (k = c*256 + kh*16 + kw): what every
    model's patch-embedding GEMM reads.  Models accept it in place of the fp32 image tensor, so one im2row pass -- plain
    (`patch_rows`) or fused with Mixup / CutMix (`mix_patch_rows`) -- serves the student, the teacher and all MultiViT
    backbones of a step.  Quacks like the image batch where host code only asks for its size and place."""

    def __init__(self, rows, B, rows_f16=None):
        self.rows, self.rows_f16, self.B = rows, rows_f16, B          # bf16 rows and / or the same values in IEEE f16
        self.shape = (B, 3, 224, 224)

    _any = property(lambda self: self.rows if self.rows is not None else self.rows_f16)
    is_cuda = property(lambda self: self._any.is_cuda)
    device = property(lambda self: self._any.device)
    dtype = torch.float32

    def of(self, dtype):
        t = self.rows_f16 if dtype == F16 else self.rows
        if t is None:
            raise L.DevitError(f"PatchRows holds no {dtype} rows: build them with patch_rows(img, dtypes=...) / "
                               "mix_patch_rows(..., dtypes=...) for every precision the models of the step run in")
        return t

    def record_stream(self, s):
        for t in (self.rows, self.rows_f16):
            if t is not None:
                t.record_stream(s)


PATCH_ROW_DTYPES = (torch.bfloat16,)      # what patch_rows / mix_patch_rows produce by default


def patch_rows(img, dtypes=None):
    """fp32 [B,3,224,224] -> PatchRows (devit_im2row_bf16); dtypes: which 16-bit copies to make (bf16 and / or f16)."""
    if isinstance(img, PatchRows):
        return img
    L.require_device(img)
    img = img.contiguous().float()
    B = img.shape[0]
    out = {}
    for dt in (dtypes or PATCH_ROW_DTYPES):
        out[dt] = rows_alloc(B * 196, 768, dt, img.device)
        call("devit_im2row_bf16", ptr(img), ptr(out[dt]), B, 3, 224, 224, 16, int(dt == F16), stream_ptr())
    return PatchRows(out.get(torch.bfloat16), B, out.get(F16))


def mix_patch_rows(img, mode, lam=1.0, box=(0, 0, 0, 0), dtypes=None):
    """Mixup (mode 1) / CutMix (mode 2, box = (y0, y1, x0, x1)) of a batch with its flip, straight to patch rows
    (devit_mix_im2row_bf16; timm Mixup mode='batch', engine.py:65-66)."""
    L.require_device(img)
    img = img.contiguous().float()
    B = img.shape[0]
    dtypes = dtypes or PATCH_ROW_DTYPES
    rows = rows_alloc(B * 196, 768, torch.bfloat16, img.device) if torch.bfloat16 in dtypes else None
    rows_h = rows_alloc(B * 196, 768, F16, img.device) if F16 in dtypes else None
    call("devit_mix_im2row_bf16", ptr(img), ptr(rows), ptr(rows_h), B, int(mode), float(lam), int(box[0]), int(box[1]),
         int(box[2]), int(box[3]), stream_ptr())
    return PatchRows(rows, B, rows_h)


def mix_targets(labels, num_classes, lam, smoothing):
    """[B, C] soft targets of the mixed batch (timm mixup_target)."""
    L.require_device(labels)
    labels = labels.contiguous().long()
    out = torch.empty((labels.shape[0], num_classes), dtype=F32, device=labels.device)
    call("devit_mix_targets", ptr(labels), ptr(out), labels.shape[0], num_classes, float(lam), float(smoothing), stream_ptr())
    return out


class PatchEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, proj_w, proj_b, cls_token, dist_token, pos_embed, w16, grad_ready):
        pre = img if isinstance(img, PatchRows) else patch_rows(img, (w16.dtype,))
        rows, B = pre.of(w16.dtype), pre.B
        dev = rows.device
        D = proj_w.shape[0]
        ntok = 2 if dist_token is not None else 1
        T = 196 + ntok
        M = B * 196
        x = torch.empty((B, T, D), dtype=F32, device=dev)
        gemm(rows, 768, 0, w16, 768, 0, pad_rows(M), D, 768, kind=L.EPI_PATCH_F32, out=x, ldc=D, bias=proj_b,
             pos=pos_embed, patch_tokens=196, extra_tokens=ntok, m_valid=M, dtype16=int(w16.dtype == F16))
        call("devit_embed_tokens", ptr(cls_token), ptr(dist_token), ptr(pos_embed), ptr(x), B, T, D, stream_ptr())
        ctx.rows, ctx.dims = rows, (B, T, D, ntok)
        ctx.params = (proj_w, proj_b, cls_token, dist_token, pos_embed)
        ctx.grad_ready = grad_ready
        return x

    @staticmethod
    def backward(ctx, dx):
        B, T, D, ntok = ctx.dims
        proj_w, proj_b, cls_token, dist_token, pos_embed = ctx.params
        dx = dx.contiguous()
        dev = dx.device
        # bf16 copy of dx with pad rows (the wgrad reduces over padded patch rows; skipped rows map past the end)
        dxb = rows_alloc(B * T, D, BF16, dev, extra=128)
        dpos = torch.empty((T, D), dtype=F32, device=dev)
        dcls = torch.empty(D, dtype=F32, device=dev)
        ddist = torch.empty(D, dtype=F32, device=dev) if ntok == 2 else None
        dbias = torch.empty(D, dtype=F32, device=dev)
        call("devit_embed_bwd", ptr(dx), B, T, D, ntok, ptr(dpos), ptr(dcls), ptr(ddist), ptr(dbias), ptr(dxb), 0,
             stream_ptr())
        grad_buf(pos_embed).view(T, D).add_(dpos)
        grad_buf(cls_token).view(D).add_(dcls)
        if ntok == 2:
            grad_buf(dist_token).view(D).add_(ddist)
        grad_buf(proj_b).add_(dbias)
        # dW[D, 768] += dx_patch^T @ rows ; reduction row r=(b,t) lives at physical row r + ntok*(r/196 + 1)
        M = B * 196
        mp = pad_rows(M)
        gemm(dxb, D, 1, ctx.rows, 768, 1, D, 768, mp, kind=L.EPI_ATOMIC_F32, out=grad_buf(proj_w), ldc=768,
             split_k=split_k_for(D, 768, mp // 64), a_group=196, a_skip=ntok)
        if ctx.grad_ready is not None:
            ctx.grad_ready([p for p in ctx.params if p is not None])
        return (None,) * 8


# ----------------------------------------------------------------------------------------------
# final norm on the cls/dist rows + classifier heads (models/de_vit.py:286-288,316-318)
# ----------------------------------------------------------------------------------------------
class HeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, norm_w, norm_b, head_w, head_b, headd_w, headd_b, ntok, eps, grad_ready):
        L.require_device(x)
        x = x.contiguous()
        B, T, D = x.shape
        rows = B * ntok
        dev = x.device
        tok = torch.empty((B, ntok, D), dtype=F32, device=dev)
        mean = torch.empty(rows, dtype=F32, device=dev)
        rstd = torch.empty(rows, dtype=F32, device=dev)
        layernorm_fwd(x.view(B * T, D), rows, D, norm_w, norm_b, eps, y_f32=tok, mean=mean, rstd=rstd, in_group=ntok,
                      in_stride=T)
        outs = [tok]
        if head_w is not None:
            Cn = head_w.shape[0]
            lo = torch.empty((B, Cn), dtype=F32, device=dev)
            sgemm_small(tok, ntok * D, 1, head_w, D, 1, head_b, lo, Cn, B, Cn, D)
            outs.append(lo)
            if ntok == 2 and headd_w is not None:
                lk = torch.empty((B, Cn), dtype=F32, device=dev)
                sgemm_small(tok[:, 1], ntok * D, 1, headd_w, D, 1, headd_b, lk, Cn, B, Cn, D)
                outs.append(lk)
        ctx.save = (x, tok, mean, rstd)
        ctx.params = (norm_w, norm_b, head_w, head_b, headd_w, headd_b)
        ctx.meta = (B, T, D, ntok, eps, grad_ready)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dtok, dlo=None, dlk=None):
        x, tok, mean, rstd = ctx.save
        norm_w, norm_b, head_w, head_b, headd_w, headd_b = ctx.params
        B, T, D, ntok, eps, grad_ready = ctx.meta
        dev = x.device
        dt = torch.zeros((B, ntok, D), dtype=F32, device=dev) if dtok is None else dtok.contiguous().clone()
        for j, (dl, w, b) in enumerate(((dlo, head_w, head_b), (dlk, headd_w, headd_b))):
            if dl is None or w is None:
                continue
            dl = dl.contiguous()
            Cn = w.shape[0]
            # dtok_j[B, D] += dl[B, C] @ w[C, D]
            sgemm_small(dl, Cn, 1, w, 1, D, None, dt[:, j], ntok * D, B, D, Cn, accumulate=True)
            # dw[C, D] += dl^T @ tok_j ; db += colsum(dl)
            sgemm_small(dl, 1, Cn, tok[:, j], 1, ntok * D, None, grad_buf(w), D, Cn, D, B, accumulate=True)
            ones = _ones1(dev)
            sgemm_small(dl, 1, Cn, ones, 0, 0, None, grad_buf(b), 1, Cn, 1, B, accumulate=True)
        dx = torch.zeros((B, T, D), dtype=F32, device=dev)
        layernorm_bwd(dt, True, x.view(B * T, D), B * ntok, D, mean, rstd, norm_w, None, dx.view(B * T, D), None, None,
                      0, grad_buf(norm_w), grad_buf(norm_b), in_group=ntok, in_stride=T)
        if grad_ready is not None:
            grad_ready([p for p in ctx.params if p is not None])
        return (dx,) + (None,) * 9


# ----------------------------------------------------------------------------------------------
# stand-alone Mlp / Attention nodes (module-level API of models/de_vit.py:21-87; the model path
# runs whole blocks through EncoderFn)
# ----------------------------------------------------------------------------------------------
def _to_rows_bf16(x):
    B, N, D = x.shape
    M = B * N
    buf = rows_alloc(M, D, BF16, x.device)
    call("devit_scale_cast_bf16", ptr(x.contiguous().float()), ptr(buf), None, 0, M, D, stream_ptr())
    return buf


class MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, w1, b1, w2, b2, w1_16, w2_16, gate, grad_enabled):
        L.require_device(x)
        B, N, D = x.shape
        M, dev, Hd, Do = B * N, x.device, w1.shape[0], w2.shape[0]
        xb = _to_rows_bf16(x)
        need = grad_enabled and (x.requires_grad or w1.requires_grad)
        h = rows_alloc(M, Hd, BF16, dev)
        h_pre = rows_alloc(M, Hd, BF16, dev) if need else None
        linear_fwd(xb, w1_16, b1, M, out=h, kind=L.EPI_GELU_BF16, colscale=gate, aux=h_pre, exact_gelu=0)
        y = torch.empty((B, N, Do), dtype=F32, device=dev)
        linear_fwd(h, w2_16, b2, M, out=y.view(M, Do), kind=L.EPI_STORE_F32)
        module.neuron_output = h[:M].view(B, N, Hd)
        ctx.s = (xb, h, h_pre, gate, w1, b1, w2, b2, w1_16, w2_16, (B, N, D))
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, h, h_pre, gate, w1, b1, w2, b2, w1_16, w2_16, (B, N, D) = ctx.s
        M, dev, Hd = B * N, dy.device, w1.shape[0]
        g = _to_rows_bf16(dy)
        dh = rows_alloc(M, Hd, BF16, dev)
        linear_dgrad(g, w2_16, M, out=dh, kind=L.EPI_DGELU_BF16, colscale=gate, aux_in=h_pre)
        linear_wgrad(g, h, grad_buf(w2), grad_buf(b2), M)
        dx = torch.empty((B, N, D), dtype=F32, device=dev)
        linear_dgrad(dh, w1_16, M, out=dx.view(M, D), kind=L.EPI_STORE_F32)
        linear_wgrad(dh, xb, grad_buf(w1), grad_buf(b1), M)
        return (dx,) + (None,) * 9


class AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, wq, bq, wp, bp_, wq16, wp16, gate, grad_enabled):
        L.require_device(x)
        B, N, D = x.shape
        M, dev, H = B * N, x.device, module.num_heads
        xb = _to_rows_bf16(x)
        need = grad_enabled and (x.requires_grad or wq.requires_grad)
        qkv = rows_alloc(M, 3 * D, BF16, dev)
        linear_fwd(xb, wq16, bq, M, out=qkv)
        o = rows_alloc(M, D, BF16, dev)
        lse = torch.empty((B, H, N), dtype=F32, device=dev) if need else None
        call("devit_attn_fwd", ptr(qkv), ptr(o), ptr(lse), ptr(gate), B, N, H, D // H, (D // H) ** -0.5, 0, stream_ptr())
        y = torch.empty((B, N, D), dtype=F32, device=dev)
        linear_fwd(o, wp16, bp_, M, out=y.view(M, D), kind=L.EPI_STORE_F32)
        module.head_output = o[:M].view(B, N, H, D // H)
        ctx.s = (xb, qkv, o, lse, gate, wq, bq, wp, bp_, wq16, wp16, (B, N, D, H))
        return y, qkv

    @staticmethod
    def backward(ctx, dy, dqkv_in):
        xb, qkv, o, lse, gate, wq, bq, wp, bp_, wq16, wp16, (B, N, D, H) = ctx.s
        M, dev = B * N, dy.device
        g = _to_rows_bf16(dy)
        do = rows_alloc(M, D, BF16, dev)
        linear_dgrad(g, wp16, M, out=do)
        linear_wgrad(g, o, grad_buf(wp), grad_buf(bp_), M)
        dqkv = rows_alloc(M, 3 * D, BF16, dev)
        call("devit_attn_bwd", ptr(qkv), ptr(o), ptr(do), ptr(lse), ptr(gate),
             ptr(dqkv_in.contiguous()) if dqkv_in is not None else None, ptr(dqkv), B, N, H, D // H, (D // H) ** -0.5,
             stream_ptr())
        dx = torch.empty((B, N, D), dtype=F32, device=dev)
        linear_dgrad(dqkv, wq16, M, out=dx.view(M, D), kind=L.EPI_STORE_F32)
        linear_wgrad(dqkv, xb, grad_buf(wq), grad_buf(bq), M)
        return (dx,) + (None,) * 9


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
class ClsDistillLossFn(torch.autograd.Function):
    """utils/losses.py:156-177 with a SoftTargetCrossEntropy base criterion; loss + gradient in one launch."""

    @staticmethod
    def forward(ctx, logits, logits_kd, teacher_logits, soft_targets, kind, alpha, tau):
        L.require_device(logits)
        logits, logits_kd = logits.contiguous().float(), logits_kd.contiguous().float()
        soft_targets = soft_targets.contiguous().float()
        tl = teacher_logits.contiguous().float() if teacher_logits is not None else None
        B, Cn = logits.shape
        loss3 = torch.empty(3, dtype=F32, device=logits.device)
        dlo, dlk = torch.empty_like(logits), torch.empty_like(logits_kd)
        call("devit_cls_distill_loss", ptr(logits), ptr(logits_kd), ptr(tl), ptr(soft_targets), B, Cn,
             {"none": 0, "soft": 1, "hard": 2}[kind], alpha, tau, ptr(loss3), ptr(dlo), ptr(dlk), stream_ptr())
        ctx.save_for_backward(dlo, dlk)
        return loss3[0]

    @staticmethod
    def backward(ctx, g):
        dlo, dlk = ctx.saved_tensors
        return dlo * g, dlk * g, None, None, None, None, None


class RelationLossFn(torch.autograd.Function):
    """q/k/v feature-relation losses (utils/losses.py:307-328) on PACKED qkv buffers.

    t_qkv: bf16 [>= B*N + 58 rows, 3*Dt], s_qkv likewise with Ds; component j lives in columns [j*D, (j+1)*D).
    Returns fp32 [3] = (q, k, v) losses.  Gradient flows to the student buffer only."""

    @staticmethod
    def forward(ctx, s_qkv, t_qkv, B, N, hd_s, hd_t):
        L.require_device(s_qkv)
        dev = s_qkv.device
        Ds, Dt = s_qkv.shape[1] // 3, t_qkv.shape[1] // 3
        assert s_qkv.shape[0] >= (B - 1) * N + 256 and t_qkv.shape[0] >= (B - 1) * N + 256, "packed qkv needs pad rows"
        losses = torch.empty(3, dtype=F32, device=dev)
        grams, stats = [], []
        for j in range(3):
            gt = torch.empty((B, 256, 256), dtype=F32, device=dev)
            gs = torch.empty((B, 256, 256), dtype=F32, device=dev)
            for buf, Dm, out in ((t_qkv, Dt, gt), (s_qkv, Ds, gs)):
                f = buf[:, j * Dm:]
                gemm(f, 3 * Dm, 0, f, 3 * Dm, 0, 256, 256, Dm, kind=L.EPI_STORE_F32, out=out, ldc=256, batch=B,
                     a_bs=N * 3 * Dm, b_bs=N * 3 * Dm, out_bs=256 * 256, dtype16=int(buf.dtype == F16))
            lse_t = torch.empty((B, N), dtype=F32, device=dev)
            lse_s = torch.empty((B, N), dtype=F32, device=dev)
            row_kl = torch.empty((B, N), dtype=F32, device=dev)
            call("devit_relation_stats", ptr(gt), ptr(gs), B, N, 256, hd_t, hd_s, ptr(lse_t), ptr(lse_s), ptr(row_kl),
                 ptr(losses[j:]), stream_ptr())
            grams.append((gt, gs))
            stats.append((lse_t, lse_s))
        ctx.grams, ctx.stats, ctx.s_qkv = grams, stats, s_qkv
        ctx.meta = (B, N, hd_s, hd_t, Ds)
        return losses

    @staticmethod
    def backward(ctx, g):
        B, N, hd_s, hd_t, Ds = ctx.meta
        s_qkv = ctx.s_qkv
        dev = s_qkv.device
        g = g.contiguous().float()
        d = torch.empty_like(s_qkv)       # rows < B*N are all written below (N rows per image x 3 components)
        S = torch.empty((B, 256, 256), dtype=BF16, device=dev)
        for j in range(3):
            gt, gs = ctx.grams[j]
            lse_t, lse_s = ctx.stats[j]
            call("devit_relation_grad", ptr(gt), ptr(gs), ptr(lse_t), ptr(lse_s), ptr(g[j:]), B, N, 256, hd_t, hd_s,
                 ptr(S), 0, stream_ptr())
            f = s_qkv[:, j * Ds:]
            # dF[b] (rows < N) = S[b] @ F[b]   (F read k-major; rows >= N of S are zero)
            gemm(S, 256, 0, f, 3 * Ds, 1, 256, Ds, 256, kind=L.EPI_STORE_BF16, out=d[:, j * Ds:], ldc=3 * Ds, batch=B,
                 a_bs=256 * 256, b_bs=N * 3 * Ds, out_bs=N * 3 * Ds, m_valid=N)
        ctx.grams = ctx.stats = None
        return d, None, None, None, None, None
