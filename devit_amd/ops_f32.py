"""Exact-fp32 parity path: the same op sequences as ops.py with fp32 activations, running on the fp32 entry points
of the C ABI (csrc/sgemm.hip + the fp32 modes of the LayerNorm / loss kernels).  Select it with
`model.precision = "f32"`; it exists so that the tests can hold the HIP path to BASELINE.json's 1e-3 bar against
the reference's fp32 CPU outputs.  It is not tuned and never benchmarked."""
import ctypes as C

import torch

from . import _lib as L
from ._lib import call, ptr, stream_ptr
from .ops import grad_buf, layernorm_bwd, layernorm_fwd

F32 = torch.float32


def sgemm(A, sam, sak, B, sbn, sbk, M, N, K, *, out, ldc, kind=L.EPI_STORE_F32, batch=1, batch_inner=1, a_bo=0, a_bi=0,
          b_bo=0, b_bi=0, c_bo=0, c_bi=0, k_group=0, k_skip=0, alpha=1.0, batch_scale=None, accumulate=False, bias=None,
          colscale=None, aux=None, aux_in=None, res=None, rowscale=None, rows_per_scale=0, pos=None, patch_tokens=0,
          extra_tokens=0, m_valid=0):
    p = lambda t: None if t is None else t.data_ptr()
    ep = L.Epilogue(kind, out.data_ptr(), ldc, p(bias), p(colscale), p(aux), p(aux_in), p(res), p(rowscale), rows_per_scale,
                    p(pos), patch_tokens, extra_tokens, 0, 0, m_valid)
    call("devit_gemm_f32", ptr(A), sam, sak, a_bo, a_bi, ptr(B), sbn, sbk, b_bo, b_bi, M, N, K, batch, batch_inner, c_bo,
         c_bi, k_group, k_skip, alpha, ptr(batch_scale), int(accumulate), C.byref(ep), stream_ptr())


def lin_fwd(x, w, bias, M, out, **kw):            # out[M, N] = x[M, K] @ w[N, K]^T
    N, K = w.shape
    sgemm(x, K, 1, w, K, 1, M, N, K, out=out, ldc=N, bias=bias, **kw)


def lin_dgrad(dy, w, M, out, **kw):               # out[M, K] = dy[M, N] @ w[N, K]
    N, K = w.shape
    sgemm(dy, N, 1, w, 1, K, M, K, N, out=out, ldc=K, **kw)


def lin_wgrad(dy, x, wg, bg, M):                  # wg[N, K] += dy[M, N]^T @ x[M, K]; bg += colsum(dy)
    N, K = wg.shape
    sgemm(dy, 1, N, x, 1, K, N, K, M, out=wg, ldc=K, accumulate=True)
    if bg is not None:
        call("devit_colsum_f32", ptr(dy), M, N, N, ptr(bg), 1, stream_ptr())


def attn_fwd(qkv, B, N, H, gate):
    D = qkv.shape[1] // 3
    dev = qkv.device
    P = torch.empty((B, H, N, N), dtype=F32, device=dev)
    sgemm(qkv, 3 * D, 1, qkv[:, D:], 3 * D, 1, N, N, 64, out=P, ldc=N, batch=B * H, batch_inner=H, a_bo=N * 3 * D, a_bi=64,
          b_bo=N * 3 * D, b_bi=64, c_bo=H * N * N, c_bi=N * N)
    call("devit_softmax_rows_f32", ptr(P), B * H * N, N, N, 0.125, None, stream_ptr())
    o = torch.empty((B * N, D), dtype=F32, device=dev)
    sgemm(P, N, 1, qkv[:, 2 * D:], 1, 3 * D, N, 64, N, out=o, ldc=D, batch=B * H, batch_inner=H, a_bo=H * N * N, a_bi=N * N,
          b_bo=N * 3 * D, b_bi=64, c_bo=N * D, c_bi=64, batch_scale=gate)
    return o, P


def attn_bwd(qkv, P, do, B, N, H, gate, dq_add):
    D = qkv.shape[1] // 3
    dev = qkv.device
    dS = torch.empty((B, H, N, N), dtype=F32, device=dev)
    kw = dict(batch=B * H, batch_inner=H)
    sgemm(do, D, 1, qkv[:, 2 * D:], 3 * D, 1, N, N, 64, out=dS, ldc=N, a_bo=N * D, a_bi=64, b_bo=N * 3 * D, b_bi=64,
          c_bo=H * N * N, c_bi=N * N, batch_scale=gate, **kw)                                   # dP = (g dO) V^T
    call("devit_softmax_bwd_rows_f32", ptr(P), ptr(dS), B * H * N, N, N, 0.125, stream_ptr())
    dqkv = dq_add.contiguous().float().clone() if dq_add is not None else torch.zeros_like(qkv)
    pn = dict(a_bo=H * N * N, a_bi=N * N, c_bo=N * 3 * D, c_bi=64, accumulate=True, **kw)
    sgemm(dS, N, 1, qkv[:, D:], 1, 3 * D, N, 64, N, out=dqkv, ldc=3 * D, b_bo=N * 3 * D, b_bi=64, **pn)         # dQ = dS K
    sgemm(dS, 1, N, qkv, 1, 3 * D, N, 64, N, out=dqkv[:, D:], ldc=3 * D, b_bo=N * 3 * D, b_bi=64, **pn)         # dK = dS^T Q
    sgemm(P, 1, N, do, 1, D, N, 64, N, out=dqkv[:, 2 * D:], ldc=3 * D, b_bo=N * D, b_bi=64, batch_scale=gate, **pn)  # dV
    return dqkv


def _scaled(dx2d, rowscale, N):
    g = torch.empty_like(dx2d)
    call("devit_scale_rows_f32", ptr(dx2d), ptr(g), ptr(rowscale), N, dx2d.shape[0], dx2d.shape[1], stream_ptr())
    return g


def block_forward(x, bp, dp, eps, need_grad, want_att):
    B, N, D = x.shape
    M, H, dev = B * N, bp.num_heads, x.device
    x2 = x.view(M, D)
    e = lambda *s: torch.empty(s, dtype=F32, device=dev)
    ln1, mean1, rstd1 = e(M, D), e(M), e(M)
    layernorm_fwd(x2, M, D, bp.n1w, bp.n1b, eps, y_f32=ln1, mean=mean1, rstd=rstd1)
    qkv = e(M, 3 * D)
    lin_fwd(ln1, bp.qkv_w, bp.qkv_b, M, qkv)
    attn_o, P = attn_fwd(qkv, B, N, H, bp.head_gate)
    x1, att = e(B, N, D), (e(M, D) if want_att else None)
    dp1, dp2 = dp if dp is not None else (None, None)
    lin_fwd(attn_o, bp.proj_w, bp.proj_b, M, x1.view(M, D), kind=L.EPI_RESIDUAL_F32, res=x2, rowscale=dp1, rows_per_scale=N,
            aux=att)
    ln2, mean2, rstd2 = e(M, D), e(M), e(M)
    layernorm_fwd(x1.view(M, D), M, D, bp.n2w, bp.n2b, eps, y_f32=ln2, mean=mean2, rstd=rstd2)
    Hd = bp.fc1_w.shape[0]
    h, h_pre = e(M, Hd), e(M, Hd)
    lin_fwd(ln2, bp.fc1_w, bp.fc1_b, M, h, kind=L.EPI_GELU_BF16, colscale=bp.neuron_gate, aux=h_pre)
    x2o = e(B, N, D)
    lin_fwd(h, bp.fc2_w, bp.fc2_b, M, x2o.view(M, D), kind=L.EPI_RESIDUAL_F32, res=x1.view(M, D), rowscale=dp2,
            rows_per_scale=N)
    if bp.module is not None:
        bp.module.mlp.neuron_output = h.view(B, N, Hd)
        bp.module.attn.head_output = attn_o.view(B, N, H, D // H)
    s = dict(x=x, ln1=ln1, mean1=mean1, rstd1=rstd1, qkv=qkv, attn_o=attn_o, P=P, x1=x1, ln2=ln2, mean2=mean2, rstd2=rstd2,
             h=h, h_pre=h_pre, dp1=dp1, dp2=dp2) if need_grad else {}
    return x2o, qkv, att, s


def block_backward(dx, s, bp, dqkv_add, datt):
    B, N, D = dx.shape
    M, H, dev = B * N, bp.num_heads, dx.device
    Hd = bp.fc1_w.shape[0]
    e = lambda *sh: torch.empty(sh, dtype=F32, device=dev)
    g2 = _scaled(dx.view(M, D), s["dp2"], N)
    dh = e(M, Hd)
    lin_dgrad(g2, bp.fc2_w, M, dh, kind=L.EPI_DGELU_BF16, colscale=bp.neuron_gate, aux_in=s["h_pre"])
    lin_wgrad(g2, s["h"], grad_buf(bp.fc2_w), grad_buf(bp.fc2_b), M)
    dln2 = e(M, D)
    lin_dgrad(dh, bp.fc1_w, M, dln2)
    lin_wgrad(dh, s["ln2"], grad_buf(bp.fc1_w), grad_buf(bp.fc1_b), M)
    dx1 = e(B, N, D)
    layernorm_bwd(dln2, True, s["x1"].view(M, D), M, D, s["mean2"], s["rstd2"], bp.n2w, dx.view(M, D), dx1.view(M, D), None,
                  None, 0, grad_buf(bp.n2w), grad_buf(bp.n2b))
    g1 = _scaled(dx1.view(M, D), s["dp1"], N)
    if datt is not None:
        g1 = g1 + datt.float().view(M, D)
    dattn = e(M, D)
    lin_dgrad(g1, bp.proj_w, M, dattn)
    lin_wgrad(g1, s["attn_o"], grad_buf(bp.proj_w), grad_buf(bp.proj_b), M)
    dqkv = attn_bwd(s["qkv"], s["P"], dattn, B, N, H, bp.head_gate, dqkv_add)
    dln1 = e(M, D)
    lin_dgrad(dqkv, bp.qkv_w, M, dln1)
    lin_wgrad(dqkv, s["ln1"], grad_buf(bp.qkv_w), grad_buf(bp.qkv_b), M)
    dx0 = e(B, N, D)
    layernorm_bwd(dln1, True, s["x"].view(M, D), M, D, s["mean1"], s["rstd1"], bp.n1w, dx1.view(M, D), dx0.view(M, D), None,
                  None, 0, grad_buf(bp.n1w), grad_buf(bp.n1b))
    return dx0


class EncoderF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg, *params):
        L.require_device(x)
        x = x.contiguous()
        need_grad = cfg.grad_enabled and (x.requires_grad or any(p.requires_grad for p in params))
        saved, qkvs, atts, encs = [], [], [], []
        for i, bp in enumerate(cfg.blocks):
            dp = cfg.dp_scales[i] if cfg.dp_scales is not None else None
            x, qkv, att, s = block_forward(x, bp, dp, cfg.eps, need_grad, cfg.want_att)
            saved.append(s)
            if cfg.want_qkv:
                qkvs.append(qkv)
            if cfg.want_att:
                atts.append(att)
            if cfg.want_enc:
                encs.append(x.clone() if i == len(cfg.blocks) - 1 else x)
        ctx.cfg, ctx.saved, ctx.need_grad = cfg, saved, need_grad
        ctx.counts = (len(qkvs), len(atts), len(encs))
        return (x,) + tuple(qkvs) + tuple(atts) + tuple(encs)

    @staticmethod
    def backward(ctx, dx, *dothers):
        cfg, saved = ctx.cfg, ctx.saved
        nq, na, ne = ctx.counts
        dqkvs, datts, dencs = dothers[:nq], dothers[nq:nq + na], dothers[nq + na:]
        nb = len(cfg.blocks)
        if dx is None:
            dx = torch.zeros_like(saved[0]["x"])
        dx = dx.contiguous()
        for i in range(nb - 1, -1, -1):
            if ne and dencs[i] is not None:
                dx = dx + dencs[i]
            dx = block_backward(dx, saved[i], cfg.blocks[i], dqkvs[i] if nq else None, datts[i] if na else None)
            saved[i] = None
            if cfg.grad_ready is not None:
                cfg.grad_ready(cfg.blocks[i].all_params())
        return (dx, None) + (None,) * (12 * nb)


class PatchEmbedF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, proj_w, proj_b, cls_token, dist_token, pos_embed, grad_ready):
        L.require_device(img)
        img = img.contiguous().float()
        B, D = img.shape[0], proj_w.shape[0]
        ntok = 2 if dist_token is not None else 1
        T, M = 196 + ntok, B * 196
        rows = torch.empty((M, 768), dtype=F32, device=img.device)
        call("devit_im2row_f32", ptr(img), ptr(rows), B, stream_ptr())
        x = torch.empty((B, T, D), dtype=F32, device=img.device)
        w2 = proj_w.detach().reshape(D, 768)
        sgemm(rows, 768, 1, w2, 768, 1, M, D, 768, out=x, ldc=D, kind=L.EPI_PATCH_F32, bias=proj_b, pos=pos_embed,
              patch_tokens=196, extra_tokens=ntok)
        call("devit_embed_tokens", ptr(cls_token), ptr(dist_token), ptr(pos_embed), ptr(x), B, T, D, stream_ptr())
        ctx.rows, ctx.dims, ctx.params, ctx.grad_ready = rows, (B, T, D, ntok), (proj_w, proj_b, cls_token, dist_token, pos_embed), grad_ready
        return x

    @staticmethod
    def backward(ctx, dx):
        B, T, D, ntok = ctx.dims
        proj_w, proj_b, cls_token, dist_token, pos_embed = ctx.params
        dx = dx.contiguous()
        dev = dx.device
        dpos, dcls, dbias = (torch.empty((T, D), dtype=F32, device=dev), torch.empty(D, dtype=F32, device=dev),
                             torch.empty(D, dtype=F32, device=dev))
        ddist = torch.empty(D, dtype=F32, device=dev) if ntok == 2 else None
        call("devit_embed_bwd", ptr(dx), B, T, D, ntok, ptr(dpos), ptr(dcls), ptr(ddist), ptr(dbias), None, 0, stream_ptr())
        grad_buf(pos_embed).view(T, D).add_(dpos)
        grad_buf(cls_token).view(D).add_(dcls)
        if ntok == 2:
            grad_buf(dist_token).view(D).add_(ddist)
        grad_buf(proj_b).add_(dbias)
        # dW[d][k] += sum_r dx[phys(r)][d] * rows[r][k],  phys(r) = r + ntok * (r / 196 + 1)
        sgemm(dx, 1, D, ctx.rows, 1, 768, D, 768, B * 196, out=grad_buf(proj_w), ldc=768, k_group=196, k_skip=ntok,
              accumulate=True)
        if ctx.grad_ready is not None:
            ctx.grad_ready([p for p in ctx.params if p is not None])
        return (None,) * 7


class RelationLossF32Fn(torch.autograd.Function):
    """utils/losses.py:307-328 on fp32 packed qkv [B*N, 3D] buffers."""

    @staticmethod
    def forward(ctx, s_qkv, t_qkv, B, N, hd_s, hd_t):
        dev = s_qkv.device
        Ds, Dt = s_qkv.shape[1] // 3, t_qkv.shape[1] // 3
        losses = torch.empty(3, dtype=F32, device=dev)
        keep = []
        for j in range(3):
            gt = torch.zeros((B, 256, 256), dtype=F32, device=dev)
            gs = torch.zeros((B, 256, 256), dtype=F32, device=dev)
            for buf, Dm, out in ((t_qkv, Dt, gt), (s_qkv, Ds, gs)):
                f = buf[:, j * Dm:]
                sgemm(f, 3 * Dm, 1, f, 3 * Dm, 1, N, N, Dm, out=out, ldc=256, batch=B, a_bo=N * 3 * Dm, b_bo=N * 3 * Dm,
                      c_bo=65536)
            lse_t, lse_s, row_kl = (torch.empty((B, N), dtype=F32, device=dev) for _ in range(3))
            call("devit_relation_stats", ptr(gt), ptr(gs), B, N, 256, hd_t, hd_s, ptr(lse_t), ptr(lse_s), ptr(row_kl),
                 ptr(losses[j:]), stream_ptr())
            keep.append((gt, gs, lse_t, lse_s))
        ctx.keep, ctx.s_qkv, ctx.meta = keep, s_qkv, (B, N, hd_s, hd_t, Ds)
        return losses

    @staticmethod
    def backward(ctx, g):
        B, N, hd_s, hd_t, Ds = ctx.meta
        s_qkv = ctx.s_qkv
        g = g.contiguous().float()
        d = torch.zeros_like(s_qkv)
        S = torch.empty((B, 256, 256), dtype=F32, device=s_qkv.device)
        for j in range(3):
            gt, gs, lse_t, lse_s = ctx.keep[j]
            call("devit_relation_grad", ptr(gt), ptr(gs), ptr(lse_t), ptr(lse_s), ptr(g[j:]), B, N, 256, hd_t, hd_s, ptr(S),
                 1, stream_ptr())
            f = s_qkv[:, j * Ds:]
            sgemm(S, 256, 1, f, 1, 3 * Ds, N, Ds, N, out=d[:, j * Ds:], ldc=3 * Ds, batch=B, a_bo=65536, b_bo=N * 3 * Ds,
                  c_bo=N * 3 * Ds)
        return d, None, None, None, None, None
