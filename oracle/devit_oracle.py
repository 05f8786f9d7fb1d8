"""CPU fp32 restatement of the DeViT hot path (functional, state_dict driven).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Parity: pinned by
tests/golden/*.npz (outputs of the reference's own modules run in the build
container by tests/golden/make_golden.py).

All citations are file:line in /root/reference.  The reference is a tree of
nn.Modules; this restatement is a set of pure functions over an ordered
``state`` dict that uses the reference's state_dict key names (SURVEY.md §8b,
"Checkpoint ABI"), so the same weights can be fed to the reference, to this
oracle and to the HIP path.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .detgen import det_array

# Geometry of the registered names on the hot path.
#   dedeit / devit : models/de_vit.py:495-513 (DeiT-S geometry: 384 / 12 / 6)
#   deit_*, vit_*  : models/deit_vit.py:457-525 (names + geometries only)
GEOMETRY = {
    "dedeit": dict(embed_dim=384, depth=12, num_heads=6, distilled=True),
    "devit": dict(embed_dim=384, depth=12, num_heads=6, distilled=False),
    "deit_tiny_patch16_224": dict(embed_dim=192, depth=12, num_heads=3, distilled=False),
    "deit_base_patch16_224": dict(embed_dim=768, depth=12, num_heads=12, distilled=False),
    "deit_tiny_distilled_patch16_224": dict(embed_dim=192, depth=12, num_heads=3, distilled=True),
    "deit_base_distilled_patch16_224": dict(embed_dim=768, depth=12, num_heads=12, distilled=True),
    # models/deit_vit.py:487-525 (vit_large is distill_sub.py:141's default --teacher-model)
    "vit_tiny_patch16_224": dict(embed_dim=192, depth=12, num_heads=3, distilled=False),
    "vit_base_patch16_224": dict(embed_dim=768, depth=12, num_heads=12, distilled=False),
    "vit_large_patch16_224": dict(embed_dim=1024, depth=24, num_heads=16, distilled=False),
}
LN_EPS = 1e-6  # models/de_vit.py:163 partial(nn.LayerNorm, eps=1e-6)


def state_keys(geom, num_classes):
    """Ordered (name, shape) list == reference state_dict order."""
    D, depth = geom["embed_dim"], geom["depth"]
    ntok = 2 if geom["distilled"] else 1
    ks = [("cls_token", (1, 1, D))]
    if geom["distilled"]:
        ks.append(("dist_token", (1, 1, D)))
    ks += [("pos_embed", (1, 196 + ntok, D)),
           ("patch_embed.proj.weight", (D, 3, 16, 16)), ("patch_embed.proj.bias", (D,))]
    for i in range(depth):
        p = f"blocks.{i}."
        ks += [(p + "norm1.weight", (D,)), (p + "norm1.bias", (D,)),
               (p + "attn.qkv.weight", (3 * D, D)), (p + "attn.qkv.bias", (3 * D,)),
               (p + "attn.proj.weight", (D, D)), (p + "attn.proj.bias", (D,)),
               (p + "norm2.weight", (D,)), (p + "norm2.bias", (D,)),
               (p + "mlp.fc1.weight", (4 * D, D)), (p + "mlp.fc1.bias", (4 * D,)),
               (p + "mlp.fc2.weight", (D, 4 * D)), (p + "mlp.fc2.bias", (D,))]
    ks += [("norm.weight", (D,)), ("norm.bias", (D,)),
           ("head.weight", (num_classes, D)), ("head.bias", (num_classes,))]
    if geom["distilled"]:
        ks += [("head_dist.weight", (num_classes, D)), ("head_dist.bias", (num_classes,))]
    return ks


def make_state(geom, num_classes, tag):
    """Deterministic weights.  Matrices std .02 like models/de_vit.py:205-216,337-369;
    biases / LN affine get small non-trivial values (the reference inits them to 0/1,
    which would hide bias and affine bugs)."""
    st = OrderedDict()
    for name, shape in state_keys(geom, num_classes):
        full = f"{tag}/{name}"
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "norm.weight":
            a = det_array(full, shape, std=0.05, mean=1.0)
        elif name.endswith(".bias"):
            a = det_array(full, shape, std=0.02)
        else:
            a = det_array(full, shape, std=0.02)
        st[name] = torch.from_numpy(a)
    return st


# ----------------------------------------------------------------------------
# model pieces
# ----------------------------------------------------------------------------
def patch_embed(st, img):
    """timm PatchEmbed used at models/de_vit.py:166-168,258: Conv2d(3,D,16,16) then
    flatten(2).transpose(1,2).  Restated as the equivalent GEMM."""
    B = img.shape[0]
    w = st["patch_embed.proj.weight"]
    D = w.shape[0]
    rows = img.reshape(B, 3, 14, 16, 14, 16).permute(0, 2, 4, 1, 3, 5).reshape(B, 196, 768)
    return rows @ w.reshape(D, 768).t() + st["patch_embed.proj.bias"]


def embed_tokens(st, img):
    """models/de_vit.py:258-264: cat(cls[, dist], patches) + pos_embed; dropout p=0."""
    x = patch_embed(st, img)
    B = x.shape[0]
    toks = [st["cls_token"].expand(B, -1, -1)]
    if "dist_token" in st:
        toks.append(st["dist_token"].expand(B, -1, -1))
    return torch.cat(toks + [x], dim=1) + st["pos_embed"]


def attention(st, pre, x, num_heads, head_gate=None):
    """models/de_vit.py:65-87.  Returns (out, (q, k, v), head_output)."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, st[pre + "qkv.weight"], st[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)   # :67
    q, k, v = qkv[0], qkv[1], qkv[2]                                     # :68
    a = (q @ k.transpose(-2, -1)) * hd ** -0.5                           # :70
    a = a.softmax(dim=-1)                                                # :71
    o = (a @ v).transpose(1, 2)                                          # :74  [B,N,H,hd]
    if head_gate is not None:                                            # :77-79
        o = o * head_gate.view(1, 1, num_heads, 1)
    head_output = o
    out = F.linear(o.reshape(B, N, C), st[pre + "proj.weight"], st[pre + "proj.bias"])  # :81-82
    return out, (q, k, v), head_output


def mlp(st, pre, x, neuron_gate=None):
    """models/de_vit.py:35-47: fc1 -> exact-erf GELU -> gate -> fc2."""
    h = F.gelu(F.linear(x, st[pre + "fc1.weight"], st[pre + "fc1.bias"]))
    if neuron_gate is not None:
        h = h * neuron_gate.view(1, 1, -1)
    return F.linear(h, st[pre + "fc2.weight"], st[pre + "fc2.bias"]), h


def block(st, i, x, num_heads, dp_scale=None, head_gate=None, neuron_gate=None):
    """models/de_vit.py:103-121.  dp_scale: optional ([B],[B]) per-sample DropPath
    multipliers floor(keep+U)/keep (models/utils/stochastic_depth.py:8-25)."""
    p = f"blocks.{i}."
    D = x.shape[-1]
    a, qkv, _ = attention(st, p + "attn.", F.layer_norm(x, (D,), st[p + "norm1.weight"],
                                                        st[p + "norm1.bias"], LN_EPS),
                          num_heads, head_gate)
    if dp_scale is not None:
        a_res = a * dp_scale[0].view(-1, 1, 1)
    else:
        a_res = a
    x = x + a_res                                                        # :114
    m, _ = mlp(st, p + "mlp.", F.layer_norm(x, (D,), st[p + "norm2.weight"],
                                            st[p + "norm2.bias"], LN_EPS), neuron_gate)
    if dp_scale is not None:
        m = m * dp_scale[1].view(-1, 1, 1)
    x = x + m                                                            # :115
    return x, qkv, a


def forward(st, geom, img, training=False, dp_scales=None, head_gates=None, neuron_gates=None):
    """models/de_vit.py:242-334 with every output flag on.

    Returns dict: 'output' (tuple in train / averaged tensor in eval for distilled
    models, :316-318), 'qkv' (list[depth] of (q,k,v) views), 'attention', 'encoder',
    'last_tokens' (post-final-LN cls/dist tokens, :288)."""
    H, depth = geom["num_heads"], geom["depth"]
    x = embed_tokens(st, img)
    enc, qkvs, atts = [x], [], []
    for i in range(depth):
        x, qkv, a = block(st, i, x, H,
                          None if dp_scales is None else dp_scales[i],
                          None if head_gates is None else head_gates[i],
                          None if neuron_gates is None else neuron_gates[i])
        enc.append(x)
        qkvs.append(qkv)
        atts.append(a)
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), st["norm.weight"], st["norm.bias"], LN_EPS)     # :286
    if geom["distilled"]:
        cls_t, dist_t = x[:, 0], x[:, 1]                                      # :288
        lo = F.linear(cls_t, st["head.weight"], st["head.bias"])              # :317
        lo_d = F.linear(dist_t, st["head_dist.weight"], st["head_dist.bias"])
        out = (lo, lo_d) if training else (lo + lo_d) / 2                     # :318
        last = (cls_t, dist_t)
    else:
        last = x[:, 0]
        out = F.linear(last, st["head.weight"], st["head.bias"])              # :328
    return {"output": out, "qkv": qkvs, "attention": atts, "encoder": enc, "last_tokens": last}


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------
def soft_target_ce(logits, soft_targets):
    """timm SoftTargetCrossEntropy (call site distill_sub.py:348): sum(-t*log_softmax).mean()."""
    return torch.sum(-soft_targets * F.log_softmax(logits, dim=-1), dim=-1).mean()


def label_smoothing_ce(x, target, smoothing=0.1):
    """utils/losses.py:10-34 (LabelSmoothingCrossEntropy): int64 labels."""
    log_prob = F.log_softmax(x, dim=-1)                                        # :23
    nll = -log_prob.gather(dim=-1, index=target.unsqueeze(1)).squeeze(1)       # :24-25
    smooth = -log_prob.mean(dim=-1)                                            # :26
    return ((1.0 - smoothing) * nll + smoothing * smooth).mean()               # :27,31


def distill_cls_loss(logits, logits_kd, teacher_logits, soft_targets, kind="hard", alpha=0.5, tau=1.0, base="soft",
                     smoothing=0.1):
    """utils/losses.py:135-177 (DistillLoss).  base: the criterion distill_sub.py:345-352 selected -- "soft"
    (SoftTargetCrossEntropy on [B, C] targets), "ls" (LabelSmoothingCrossEntropy(smoothing)) or "ce" (nn.CrossEntropyLoss),
    the last two on int64 labels passed as `soft_targets`."""
    if base == "soft":
        base = soft_target_ce(logits, soft_targets)                           # :171
    elif base == "ls":
        base = label_smoothing_ce(logits, soft_targets, smoothing)
    else:
        base = F.cross_entropy(logits, soft_targets)
    if kind == "none":
        return base
    if kind == "hard":
        dist = F.cross_entropy(logits_kd, teacher_logits.argmax(dim=1))       # :153
    else:
        a = F.log_softmax(logits_kd / tau, dim=1)
        b = F.log_softmax(teacher_logits / tau, dim=1)
        dist = torch.sum(b.exp() * (b - a)) * (tau * tau) / logits_kd.numel()  # :140-148
    return base * (1 - alpha) + dist * alpha                                   # :176


def feature_relation_loss(teacher_feature, student_feature):
    """utils/losses.py:307-328.  Inputs [B, H, N, hd] (strided views are fine)."""
    B, _, N, thd = teacher_feature.shape
    shd = student_feature.shape[-1]
    tf = teacher_feature.permute(0, 2, 1, 3).reshape(B, N, -1)                 # :313-314
    sf = student_feature.permute(0, 2, 1, 3).reshape(B, N, -1)                 # :315-316
    t = F.log_softmax(tf @ tf.transpose(-1, -2) / math.sqrt(thd), dim=-1)      # :318-320
    s = F.log_softmax(sf @ sf.transpose(-1, -2) / math.sqrt(shd), dim=-1)      # :322-324
    return torch.sum(t.exp() * (t - s)) / B                                    # :309,326 batchmean


def distill_step(st_s, geom_s, st_t, geom_t, img, soft_targets, gama=(0.2, 0.1, 0.3),
                 kind="hard", alpha=0.5, tau=1.0, dp_scales=None, head_gates=None, neuron_gates=None):
    """engine.py:68-106 restated: student train-mode forward, teacher eval forward
    (no grad), DEKD cls loss + q/k/v relation losses on layer depth//2-1."""
    so = forward(st_s, geom_s, img, training=True, dp_scales=dp_scales,
                 head_gates=head_gates, neuron_gates=neuron_gates)
    with torch.no_grad():
        to = forward(st_t, geom_t, img, training=False)
    cls_loss = distill_cls_loss(so["output"][0], so["output"][1], to["output"], soft_targets,
                                kind, alpha, tau)                              # :79
    ls, lt = geom_s["depth"], geom_t["depth"]
    s_qkv, t_qkv = so["qkv"][ls // 2 - 1], to["qkv"][lt // 2 - 1]              # :91-92
    q_loss, k_loss, v_loss = [feature_relation_loss(tv, sv) / ls for sv, tv in zip(s_qkv, t_qkv)]  # :95-104
    loss = cls_loss + gama[0] * q_loss + gama[1] * k_loss + gama[2] * v_loss   # :105-106
    return {"loss": loss, "cls_loss": cls_loss, "q_loss": q_loss, "k_loss": k_loss, "v_loss": v_loss,
            "student": so, "teacher": to}


# ----------------------------------------------------------------------------
# ensemble stage (models/ensemble_models.py, utils/losses.py:180-244)
# ----------------------------------------------------------------------------
def make_ens_state(tag="ENS", sum_dim=1536, teacher_size=768, num_class=100):
    """EnsMLP weights in the reference's registration order (models/ensemble_models.py:55-63)."""
    shapes = [("cls_mlp.weight", (teacher_size, sum_dim)), ("cls_mlp.bias", (teacher_size,)),
              ("cls_classifier.weight", (num_class, teacher_size)), ("cls_classifier.bias", (num_class,)),
              ("dist_mlp.weight", (teacher_size, sum_dim)), ("dist_mlp.bias", (teacher_size,)),
              ("dist_classifier.weight", (num_class, teacher_size)), ("dist_classifier.bias", (num_class,))]
    return OrderedDict((k, torch.from_numpy(det_array(f"{tag}/{k}", sh, std=0.02))) for k, sh in shapes)


def ens_forward(sub_states, geom, ens, img, training=False):
    """MultiViT.forward (:32-40) + EnsMLP.forward (:65-90) for distilled ('deit') sub-models.
    Returns ((cls_token, dist_token), logits)."""
    feats = [forward(st, geom, img, training=training)["last_tokens"] for st in sub_states]
    cls_cat = torch.stack([f[0] for f in feats], 1).reshape(img.shape[0], -1)      # :77
    dist_cat = torch.stack([f[1] for f in feats], 1).reshape(img.shape[0], -1)     # :78
    ct = F.linear(cls_cat, ens["cls_mlp.weight"], ens["cls_mlp.bias"])             # :80-81
    dt = F.linear(dist_cat, ens["dist_mlp.weight"], ens["dist_mlp.bias"])
    logits = (F.linear(ct, ens["cls_classifier.weight"], ens["cls_classifier.bias"]) +
              F.linear(dt, ens["dist_classifier.weight"], ens["dist_classifier.bias"])) / 2    # :84-86
    return (ct, dt), logits


def ens_loss(tokens, logits, teacher_out, soft_targets, kind="hard", alpha=0.5, tau=1.0):
    """EnsLoss.forward, 'deit' branch (utils/losses.py:233-244) -> (token_loss, cls_loss)."""
    tea_cls, tea_dist = teacher_out["last_tokens"]
    cls_loss = distill_cls_loss(logits, logits, teacher_out["output"], soft_targets, kind, alpha, tau)
    token_loss = F.mse_loss(tokens[0], tea_cls) + F.mse_loss(tokens[1], tea_dist)
    return token_loss, cls_loss
