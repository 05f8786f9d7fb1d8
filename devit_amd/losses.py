"""DEKD losses with the reference's API (utils/losses.py), computed by libdevit_hip.so.

  DistillLoss(base_criterion, distillation_type, alpha, tau)(outputs, teacher_outputs, labels)   :122-177
  feature_relation_loss(teacher_feature, student_feature)                                        :307-328
  relation_losses_packed(student_qkv, teacher_qkv)   -- the three q/k/v losses of engine.py:95 in one node
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import ops


class SoftTargetCrossEntropy(nn.Module):
    """timm.loss.SoftTargetCrossEntropy (distill_sub.py:348): sum(-t * log_softmax(x)).mean().
    A marker class: DistillLoss fuses it with the distillation term in one kernel."""

    def forward(self, x, target):
        zero = torch.zeros_like(x)
        return ops.ClsDistillLossFn.apply(x, x, zero, target, "none", 0.0, 1.0)


def smoothed_one_hot(labels, num_classes, smoothing):
    """[B, C] fp32 rows with 1 - smoothing + smoothing / C on the label and smoothing / C elsewhere: soft-target CE on these rows
    == LabelSmoothingCrossEntropy(smoothing) (utils/losses.py:25-31: confidence * nll + smoothing * mean(-log p))."""
    off = smoothing / num_classes
    return torch.full((labels.shape[0], num_classes), off, dtype=torch.float32, device=labels.device) \
        .scatter_(1, labels.long()[:, None], 1.0 - smoothing + off)


class DistillLoss(nn.Module):
    """utils/losses.py:122-177 with any of the three base criteria distill_sub.py:345-352 can pick: SoftTargetCrossEntropy
    (mixup on: `labels` are the soft [B, C] targets of engine.py:66), LabelSmoothingCrossEntropy(smoothing) (mixup off,
    --smoothing > 0) and nn.CrossEntropyLoss (both off) on int64 labels -- the last two are the soft-target kernel on
    smoothed / plain one-hot rows."""

    def __init__(self, base_criterion, distillation_type, alpha, tau):
        super().__init__()
        assert distillation_type in ['none', 'soft', 'hard']
        if not isinstance(base_criterion, (SoftTargetCrossEntropy, nn.CrossEntropyLoss, LabelSmoothingCrossEntropy)):
            raise NotImplementedError("DistillLoss is fused for the base criteria distill_sub.py:345-352 selects: "
                                      "SoftTargetCrossEntropy, LabelSmoothingCrossEntropy, nn.CrossEntropyLoss")
        if isinstance(base_criterion, nn.CrossEntropyLoss) and (
                base_criterion.weight is not None or base_criterion.label_smoothing or base_criterion.reduction != "mean"
                or base_criterion.ignore_index != -100):
            raise NotImplementedError("DistillLoss: nn.CrossEntropyLoss() with default arguments only (distill_sub.py:352)")
        self.base_criterion = base_criterion
        self.distillation_type = distillation_type
        self.alpha = alpha
        self.tau = tau

    def forward(self, outputs, teacher_outputs, labels):
        if not isinstance(outputs, torch.Tensor):
            outputs, outputs_kd = outputs         # (cls head, dist head), utils/losses.py:164-167
        else:
            outputs_kd = outputs
        hard = labels.dtype in (torch.int64, torch.int32)
        if isinstance(self.base_criterion, LabelSmoothingCrossEntropy):
            if not hard:
                raise ValueError("LabelSmoothingCrossEntropy takes int64 class labels (utils/losses.py:23 gathers by them)")
            labels = smoothed_one_hot(labels, outputs.shape[1], self.base_criterion.smoothing)
        elif hard:
            labels = smoothed_one_hot(labels, outputs.shape[1], 0.0)
        return ops.ClsDistillLossFn.apply(outputs, outputs_kd, teacher_outputs, labels, self.distillation_type,
                                          float(self.alpha), float(self.tau))


def _packed_of(t):
    p = getattr(t, "_devit_packed", None)
    return p


def relation_losses_packed(student_qkv, teacher_qkv):
    """(q_loss, k_loss, v_loss) for one layer pair.  Arguments are the (q, k, v) tuples returned by the models
    with output_qkv=True (strided views of the packed qkv GEMM output)."""
    sp, tp = _packed_of(student_qkv[0]), _packed_of(teacher_qkv[0])
    if sp is None or tp is None:
        return tuple(feature_relation_loss(tv, sv) for sv, tv in zip(student_qkv, teacher_qkv))
    (s_buf, B, N, Hs), (t_buf, Bt, Nt, Ht) = sp, tp
    assert (B, N) == (Bt, Nt)
    hd_s, hd_t = s_buf.shape[1] // (3 * Hs), t_buf.shape[1] // (3 * Ht)
    if s_buf.dtype == torch.float32:          # exact-fp32 parity path
        from . import ops_f32
        losses = ops_f32.RelationLossF32Fn.apply(s_buf, t_buf.detach().float(), B, N, hd_s, hd_t)
        return losses[0], losses[1], losses[2]
    if t_buf.dtype == torch.float32:          # (an fp32-path teacher under a 16-bit student: the generic per-feature form repacks it)
        return tuple(feature_relation_loss(tv, sv) for sv, tv in zip(student_qkv, teacher_qkv))
    losses = ops.RelationLossFn.apply(s_buf, t_buf.detach(), B, N, hd_s, hd_t)
    return losses[0], losses[1], losses[2]


def relation_losses_vector(student_qkv, teacher_qkv):
    """The same three losses as ONE tensor [q, k, v] (or None when the inputs are not the packed qkv views of the HIP
    models).  engine.distill_forward forms the total on this vector: three `select`s of it would cost three
    select_backward kernels plus two accumulations in backward just to rebuild the 3-vector gradient."""
    sp, tp = _packed_of(student_qkv[0]), _packed_of(teacher_qkv[0])
    if sp is None or tp is None:
        return None
    (s_buf, B, N, Hs), (t_buf, Bt, Nt, Ht) = sp, tp
    assert (B, N) == (Bt, Nt)
    hd_s, hd_t = s_buf.shape[1] // (3 * Hs), t_buf.shape[1] // (3 * Ht)
    if s_buf.dtype == torch.float32:          # exact-fp32 parity path
        from . import ops_f32
        return ops_f32.RelationLossF32Fn.apply(s_buf, t_buf.detach().float(), B, N, hd_s, hd_t)
    if t_buf.dtype == torch.float32:
        return None                           # (the caller falls back to relation_losses_packed's generic form)
    return ops.RelationLossFn.apply(s_buf, t_buf.detach(), B, N, hd_s, hd_t)


def _repack(feature):
    """[B, H, N, hd] (any strides) -> packed bf16 [pad(B*N) + 128, 3*H*hd] with the feature in every component."""
    B, H, N, hd = feature.shape
    D = H * hd
    buf = ops.rows_alloc(B * N, 3 * D, torch.bfloat16, feature.device, extra=128)
    f = feature.permute(0, 2, 1, 3).reshape(B * N, D)
    return buf, f, (B, N, H, hd, D)


class _RepackFn(torch.autograd.Function):
    """Differentiable copy of one [B,H,N,hd] feature into component 0 of a zeroed packed buffer."""

    @staticmethod
    def forward(ctx, feature):
        buf, f, meta = _repack(feature)
        buf[: f.shape[0], : meta[4]] = f.to(torch.bfloat16)
        buf[: f.shape[0], meta[4]:].zero_()
        ctx.meta = meta
        return buf

    @staticmethod
    def backward(ctx, g):
        B, N, H, hd, D = ctx.meta
        return g[: B * N, :D].reshape(B, N, H, hd).permute(0, 2, 1, 3).float()


def feature_relation_loss(teacher_feature, student_feature):
    """utils/losses.py:307-328 (note the argument order).  Generic entry: accepts any [B, H, N, hd] tensors;
    the engine uses relation_losses_packed, which reads the qkv GEMM output in place."""
    L.require_device(student_feature)
    B, Hs, N, hd_s = student_feature.shape
    _, Ht, _, hd_t = teacher_feature.shape
    s_buf = _RepackFn.apply(student_feature)
    with torch.no_grad():
        t_buf = _RepackFn.apply(teacher_feature)
    losses = ops.RelationLossFn.apply(s_buf, t_buf, B, N, hd_s, hd_t)
    return losses[0]


class DistillationLoss(nn.Module):
    """utils/losses.py:44-118 (DeiT's criterion, used by train_subdata.py:423-425): base criterion on the class-token
    logits plus hard / soft distillation of the distillation-token logits against a teacher that is run INSIDE the
    criterion, under no_grad, on the same inputs (:104-106); `(1 - alpha) * base + alpha * distill` (:115).
    distillation_type 'none' returns the base loss (:100-101).  The distill_token variant (token MSE against the
    teacher's last tokens, :107-112) needs --distillation-token / resize_dim, which is not on the DeViT path."""

    def __init__(self, base_criterion, teacher_model, distillation_type, alpha, tau, distill_token=False):
        super().__init__()
        assert distillation_type in ['none', 'soft', 'hard']
        if distill_token:
            raise NotImplementedError("DistillationLoss(distill_token=True) is not built: utils/losses.py:106-108 unpacks the teacher's "
                                      "return value as (token, logits), which the repository's own model never returns (it returns a "
                                      "dict) -- DESIGN.md section 9")
        self.base_criterion, self.teacher_model = base_criterion, teacher_model
        self.distillation_type, self.alpha, self.tau = distillation_type, alpha, tau
        # LabelSmoothingCrossEntropy (train_subdata.py:411-416 without mixup): DistillLoss smooths the one-hot rows itself
        self._loss = DistillLoss(base_criterion, distillation_type, alpha, tau)

    def forward(self, inputs, outputs, labels, token_outputs=None):
        if self.distillation_type == 'none':
            logits = outputs if isinstance(outputs, torch.Tensor) else outputs[0]
            return self._loss((logits, logits), None, labels)
        if isinstance(outputs, torch.Tensor):
            raise ValueError("DistillationLoss: the model must return (outputs, outputs_kd) when distilling")
        from .de_vit import lean_tail
        with torch.no_grad(), lean_tail(self.teacher_model):       # only the teacher's logits are read
            teacher_outputs = self.teacher_model(inputs)
        return self._loss(outputs, teacher_outputs, labels)


class LabelSmoothingCrossEntropy(nn.Module):
    """utils/losses.py:10-34; expressed through the fused kernel with smoothed one-hot targets."""

    def __init__(self, smoothing=0.1):
        super().__init__()
        assert smoothing < 1.0
        self.smoothing = smoothing
        self.confidence = 1. - smoothing

    def forward(self, x, target):
        soft = smoothed_one_hot(target, x.shape[-1], self.smoothing)
        return ops.ClsDistillLossFn.apply(x, x, torch.zeros_like(x), soft, "none", 0.0, 1.0)


class _TokenMseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        from ._lib import call, ptr, stream_ptr
        L.require_device(a)
        a, b = a.contiguous().float(), b.contiguous().float()
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        da = torch.empty_like(a)
        call("devit_token_mse", ptr(a), ptr(b), a.numel(), ptr(loss), ptr(da), 0, stream_ptr())
        ctx.save_for_backward(da)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        da, = ctx.saved_tensors
        return da * g, None


class EnsLoss(nn.Module):
    """utils/losses.py:180-244: (token_loss, cls_loss) of the ensemble stage.  The teacher forward runs inside, under
    no_grad, with distill_token=True (:222-227)."""

    def __init__(self, base_criterion, teacher_model, model, distillation_type, alpha, tau, loss_type='mse'):
        super().__init__()
        if loss_type != 'mse':
            raise NotImplementedError("EnsLoss: only loss_type='mse' is on the DeViT path (ensemble.py default)")
        self.base_criterion, self.teacher_model, self.model = base_criterion, teacher_model, model
        self.distillation_type, self.alpha, self.tau = distillation_type, alpha, tau
        self._cls = DistillLoss(base_criterion, distillation_type, alpha, tau)     # raises for a base criterion it does not fuse

    def forward(self, inputs, stu_outputs, labels):
        if self.distillation_type == 'none':
            return self._cls(stu_outputs, None, labels)
        from .de_vit import lean_tail
        with torch.no_grad(), lean_tail(self.teacher_model):       # logits and the last class / distillation tokens are read
            tea = self.teacher_model(inputs, distill_token=True)
        tokens, stu_logits = stu_outputs
        cls_loss = self._cls(stu_logits, tea['output'], labels)          # (1-a) base + a distill, :236-237
        if 'vit' in self.model:
            return _TokenMseFn.apply(tokens, tea['last_tokens']), cls_loss
        cls_token, dist_token = tokens
        tea_token, tea_token_dist = tea['last_tokens']
        return _TokenMseFn.apply(cls_token, tea_token) + _TokenMseFn.apply(dist_token, tea_token_dist), cls_loss
