"""Collaborative-inference models of DeViT (models/ensemble_models.py): `MultiViT` = N shrunk/distilled sub-model
backbones run on the same batch, `EnsMLP` = token concat -> Linear -> classifier, cls/dist logits averaged.
Backbones run on the HIP block kernels; the small fusion Linears (M = batch) run on the fp32 GEMM entry point."""
import torch
import torch.nn as nn

from . import _lib as L
from .registry import create_model


class MultiViT(nn.Module):
    """models/ensemble_models.py:13-40."""

    def __init__(self, model='dedeit', drop=0, drop_path=0.1, num_classes_list=[25, 25, 25, 25], num_div=4):
        super().__init__()
        self.model = model
        assert len(num_classes_list) == num_div, 'num of classes is not match num of sub-models'
        self.backbones = nn.ModuleList([])
        for i, num_class in enumerate(num_classes_list):
            self.backbones.append(create_model(model_name=self.model, num_classes=int(num_class), drop_rate=drop,
                                               drop_path_rate=drop_path, drop_block_rate=None))
            del self.backbones[i].head
            self.backbones[i].head = nn.Identity()          # attribute must exist for forward_features' head plumbing
            if 'deit' in self.model:
                del self.backbones[i].head_dist
                self.backbones[i].head_dist = nn.Identity()

    def forward(self, x):
        # the backbones read the same batch: cut it into bf16 patch rows ONCE (the reference runs four patch-embedding
        # convolutions over the same images, :33-39)
        from . import ops
        if x.is_cuda and all(getattr(m, "precision", "bf16") == "bf16" for m in self.backbones):
            x = ops.patch_rows(x)
        # only the class / distillation tokens of every backbone are read (:36-39): their last blocks run on the token rows
        from .de_vit import lean_tail
        with lean_tail(*self.backbones):
            feats = [m.forward_features(x) for m in self.backbones]
        if 'vit' in self.model:
            return [f['output'] for f in feats]
        return [f['output'][0] for f in feats], [f['output'][1] for f in feats]


class _LinearF32Fn(torch.autograd.Function):
    """y = x W^T + b on the exact-fp32 GEMM (small M: batch rows)."""

    @staticmethod
    def forward(ctx, x, w, b):
        from . import ops_f32
        L.require_device(x)
        x = x.contiguous().float()
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        ops_f32.sgemm(x, K, 1, w, K, 1, M, N, K, out=y, ldc=N, bias=b)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops_f32
        from ._lib import call, ptr, stream_ptr
        x, w = ctx.saved_tensors
        dy = dy.contiguous().float()
        M, K = x.shape
        N = w.shape[0]
        dx = torch.empty_like(x)
        ops_f32.sgemm(dy, N, 1, w, 1, K, M, K, N, out=dx, ldc=K)
        dw = torch.empty_like(w)
        ops_f32.sgemm(dy, 1, N, x, 1, K, N, K, M, out=dw, ldc=K)
        db = None
        if ctx.has_bias:
            db = torch.empty(N, dtype=torch.float32, device=x.device)
            call("devit_colsum_f32", ptr(dy), M, N, N, ptr(db), 0, stream_ptr())
        return dx, dw, db


def linear_f32(x, lin):
    return _LinearF32Fn.apply(x, lin.weight, lin.bias)


class EnsMLP(nn.Module):
    """models/ensemble_models.py:43-90 (same parameter names / registration order)."""

    def __init__(self, model='dedeit', num_class=100, sub_size=384, num_classes_list=[25, 25, 25, 25], teacher_size=None):
        super().__init__()
        self.model, self.sub_size, self.teacher_size, self.num_classes = model, sub_size, teacher_size, num_class
        self.sum_feature_dim = self.sub_size * len(num_classes_list)
        if self.teacher_size is None:
            self.cls_classifier = nn.Linear(self.sum_feature_dim, self.num_classes)
            if 'deit' in self.model:
                self.dist_classifier = nn.Linear(self.sum_feature_dim, self.num_classes)
        else:
            self.cls_mlp = nn.Linear(self.sum_feature_dim, self.teacher_size)
            self.cls_classifier = nn.Linear(self.teacher_size, self.num_classes)
            if 'deit' in self.model:
                self.dist_mlp = nn.Linear(self.sum_feature_dim, self.teacher_size)
                self.dist_classifier = nn.Linear(self.teacher_size, self.num_classes)

    def forward(self, x, distill=False):
        if 'vit' in self.model:
            ens_cls = torch.stack(x, 1).view(x[0].shape[0], -1)
            if self.teacher_size is not None:
                ens_cls = linear_f32(ens_cls, self.cls_mlp)
            ens_token = ens_cls
            logits = linear_f32(ens_cls, self.cls_classifier)
        else:
            cls_list, dist_list = x
            ens_cls = torch.stack(cls_list, 1).view(cls_list[0].shape[0], -1)
            ens_dist = torch.stack(dist_list, 1).view(dist_list[0].shape[0], -1)
            if self.teacher_size is not None:
                ens_cls, ens_dist = linear_f32(ens_cls, self.cls_mlp), linear_f32(ens_dist, self.dist_mlp)
            ens_token = (ens_cls, ens_dist)
            logits = (linear_f32(ens_cls, self.cls_classifier) + linear_f32(ens_dist, self.dist_classifier)) / 2
        if distill and self.training and self.teacher_size is not None:
            return ens_token, logits
        return logits


def load_sub_checkpoints(model: MultiViT, state_dicts):
    """ensemble.py:192-200,229-238: positional copy of every sub-model checkpoint minus its LAST 4 (deit) / 2 (vit)
    tensors -- the classifier heads -- into MultiViT.state_dict(), in registration order."""
    multi = model.state_dict()
    keys = list(multi.keys())
    drop = 2 if 'vit' in model.model else 4
    for i, sd in enumerate(state_dicts):
        src = list(sd.keys())
        n = len(src) - drop
        for j in range(n):
            multi[keys[i * n + j]] = sd[src[j]]
    model.load_state_dict(multi)
    return model
