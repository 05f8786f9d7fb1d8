"""MI355X-native mirror of the reference's gated ViT/DeiT module surface (models/de_vit.py).

Same class names (`Mlp`, `Attention`, `Block`, `VisionTransformer`), constructor arguments, state_dict
keys/order/shapes (155 tensors for distilled models), forward() return-type matrix
(models/de_vit.py:316-334) and shrink contract (`gate`, `neuron_output`, `head_output`,
`hidden_features`, `num_heads`; core/imp_rank.py).  The arithmetic runs in libdevit_hip.so only: modules
own fp32 master parameters, keep bf16 copies for the MFMA GEMMs, and call devit_amd.ops.  A CPU tensor
raises (no fallback); the CPU restatement used for parity lives in oracle/.

The nn.Linear / nn.LayerNorm / nn.Conv2d children are parameter containers (so initialisation and
checkpoint keys equal the reference's); their own forward() is never called.
"""
import contextlib
import math
import os
from functools import partial

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .registry import register_model

__all__ = ["Mlp", "Attention", "Block", "PatchEmbed", "VisionTransformer", "model_config", "dedeit", "devit"]


def _w16(lin, f16=False):
    """16-bit GEMM copy of an nn.Linear / Conv2d weight (bf16; IEEE f16 for precision="f16"), refreshed when the fp32
    master changed.  (Flat optimizers that rewrite master + bf16 through the C ABI keep both in sync themselves.)"""
    w = lin.weight
    attr = "_w16h" if f16 else "_w16"
    c = getattr(lin, attr, None)
    if c is None or c[0] != w._version or c[1].device != w.device or c[2] != w.data_ptr():
        buf = c[1] if (c is not None and c[1].device == w.device and c[1].numel() == w.numel()) else None
        t = ops.cast_bf16(w.detach().reshape(w.shape[0], -1), buf, f16=f16)
        setattr(lin, attr, (w._version, t, w.data_ptr()))
        return t
    return c[1]


def _w16t(lin, w16):
    """K-major bf16 copy [in][out] of an nn.Linear's bf16 GEMM copy `w16` ([out][in]) for the full-row 256x384 GEMM (csrc/gemm.hip: N == 384
    outputs), re-derived whenever `w16` was re-made; in-place rewrites of `w16` (fused optimizer, FlatParams.refresh_bf16) re-derive it through
    ddp.FlatParams.refresh_kmajor()."""
    c = lin.__dict__.get("_w16t")
    if c is None or c[0] != w16.data_ptr() or c[1] != lin.weight._version or c[2].device != w16.device:
        buf = c[2] if (c is not None and c[2].device == w16.device and c[2].numel() == w16.numel()) else \
            torch.empty((w16.shape[1], w16.shape[0]), dtype=w16.dtype, device=w16.device)
        ops.transpose16(w16, buf)
        lin._w16t = (w16.data_ptr(), lin.weight._version, buf, w16)
        return buf
    return c[2]


class _Gated:
    """`gate` is a plain CPU float tensor attribute in the reference (de_vit.py:33,63), re-uploaded on every
    forward (:42,:78).  Here the device copy is cached and all-ones gates are skipped."""

    def _init_gate(self, n):
        self._gate = torch.ones(n)
        self._gate_dev = None
        self._gate_ones = True
        self._gate_version = 0        # bumped by every assignment: cache keys use it, never id() (ids of freed tensors are reused)
        self._gate_seen = (0, self._gate._version)

    @property
    def gate(self):
        return self._gate

    @gate.setter
    def gate(self, value):
        self._gate = value
        self._gate_dev = None
        self._gate_ones = None        # unknown until the next forward looks (the shrink code ASSIGNS gates, imp_rank.py:65-71)
        self._gate_version += 1

    def gate_key(self):
        """Changes whenever the gate was assigned or written in place."""
        return (self._gate_version, self._gate._version)

    def gate_on(self, device):
        g = self._gate
        if self._gate_seen != self.gate_key():      # assigned, or modified in place (m.gate[j] = 0), since the last look
            self._gate_seen = self.gate_key()
            self._gate_dev, self._gate_ones = None, None
        if self._gate_ones is None:
            self._gate_ones = bool((g == 1).all())
        if self._gate_ones:
            return None
        if self._gate_dev is None or self._gate_dev.device != device:
            self._gate_dev = g.detach().float().to(device).contiguous()
        return self._gate_dev


class Mlp(nn.Module, _Gated):
    """fc1 -> exact GELU -> neuron gate -> fc2 (models/de_vit.py:21-47)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.hidden_features = hidden_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        if drop != 0.:
            raise NotImplementedError("dropout p > 0 is not on the DeViT hot path (distill_sub.py --drop 0.0)")
        self._init_gate(hidden_features)

    def forward(self, x):
        return _standalone_mlp(self, x)


class Attention(nn.Module, _Gated):
    """qkv -> softmax(q k^T / sqrt(hd)) v -> head gate -> proj (models/de_vit.py:50-87)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        if attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError("dropout p > 0 is not on the DeViT hot path")
        if head_dim != 64:
            raise NotImplementedError("the fused attention kernel is built for head_dim == 64 (all DeiT/ViT-16 models)")
        self._init_gate(num_heads)

    def forward(self, x, output_qkv=False):
        return _standalone_attention(self, x, output_qkv)


def qkv_views(qkv_packed, B, N, H):
    """(q, k, v) strided views [B, H, N, hd] of the packed qkv GEMM output (de_vit.py:67-68)."""
    hd = qkv_packed.shape[1] // (3 * H)
    v = qkv_packed[: B * N].view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, vv = v[0], v[1], v[2]
    for t in (q, k, vv):
        t._devit_packed = (qkv_packed, B, N, H)   # lets losses.relation_losses_packed skip the re-pack
    return q, k, vv


class Block(nn.Module):
    """models/de_vit.py:90-121.  Runs as one fused autograd node (ops.EncoderFn)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    @property
    def drop_prob(self):
        return self.drop_path.drop_prob if isinstance(self.drop_path, DropPath) else 0.

    def block_params(self, device, f16=False):
        """Parameter view of this block for ops.EncoderFn, cached between calls: rebuilt when a weight was rewritten
        (version counter / storage), a gate was assigned, the block was (un)compacted or switched train / eval."""
        c = getattr(self, "_compact", None)
        a, m_ = self.attn, self.mlp
        wk = "_w16h" if f16 else "_w16"
        def w16_ptr(lin):      # the cached BlockParams holds this tensor, so its address cannot be handed out again meanwhile
            t = lin.__dict__.get(wk)
            return t[1].data_ptr() if t is not None else 0
        def key():
            return (device, self.training, self.drop_prob, f16, getattr(self, "_compact_version", 0), c is not None,
                    a.gate_key(), m_.gate_key(), w16_ptr(a.qkv), w16_ptr(a.proj), w16_ptr(m_.fc1), w16_ptr(m_.fc2),
                    a.qkv.weight._version, a.proj.weight._version, m_.fc1.weight._version, m_.fc2.weight._version,
                    a.qkv.weight.data_ptr(), a.proj.weight.data_ptr(), m_.fc1.weight.data_ptr(), m_.fc2.weight.data_ptr())
        cached = getattr(self, "_bp_cache", None)
        if cached is not None and cached[0] == key():
            return cached[1]
        bp = self._build_block_params(device, f16)
        self._bp_cache = (key(), bp)         # (the build may have made the 16-bit copies: key taken after it)
        return bp

    def _build_block_params(self, device, f16=False):
        bp = ops.BlockParams()
        bp.compacted = False
        c = getattr(self, "_compact", None)
        if f16 and c is not None:
            raise L.DevitError('precision="f16" and shrink.compact() are not combined: uncompact the model first')
        if c is not None:            # physically shrunk weights (shrink.compact): gates folded in, masked units gone
            bp.n1w, bp.n1b, bp.n2w, bp.n2b = self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias
            bp.qkv_w = bp.proj_w = bp.fc1_w = bp.fc2_w = None
            bp.qkv_b, bp.proj_b, bp.fc1_b, bp.fc2_b = c["qkv_b"], self.attn.proj.bias, c["fc1_b"], self.mlp.fc2.bias
            bp.qkv_w16, bp.proj_w16, bp.fc1_w16, bp.fc2_w16 = c["qkv_w16"], c["proj_w16"], c["fc1_w16"], c["fc2_w16"]
            bp.fc2_w16t = c.get("fc2_w16t")
            bp.num_heads, bp.neuron_gate = c["num_heads"], None
            bp.head_gate = None if c.get("heads_compacted", True) else self.attn.gate_on(device)
            bp.dp_prob, bp.module, bp.compacted = (self.drop_prob if self.training else 0.), self, True
            bp.masters = bp.finish = None
            if c.get("trainable"):   # training through the compacted block: compact weight gradients -> the masters'
                from . import shrink
                shrink.attach_training(self, bp, c)
            return bp
        bp.n1w, bp.n1b = self.norm1.weight, self.norm1.bias
        bp.qkv_w, bp.qkv_b = self.attn.qkv.weight, self.attn.qkv.bias
        bp.proj_w, bp.proj_b = self.attn.proj.weight, self.attn.proj.bias
        bp.n2w, bp.n2b = self.norm2.weight, self.norm2.bias
        bp.fc1_w, bp.fc1_b = self.mlp.fc1.weight, self.mlp.fc1.bias
        bp.fc2_w, bp.fc2_b = self.mlp.fc2.weight, self.mlp.fc2.bias
        bp.qkv_w16, bp.proj_w16 = _w16(self.attn.qkv, f16), _w16(self.attn.proj, f16)     # (unused by the fp32 parity path)
        bp.fc1_w16, bp.fc2_w16 = _w16(self.mlp.fc1, f16), _w16(self.mlp.fc2, f16)
        # fc2 of a 384-wide model runs on the full-row GEMM, which reads its weight k-major
        bp.fc2_w16t = _w16t(self.mlp.fc2, bp.fc2_w16) if (not f16 and bp.fc2_w16.is_cuda and bp.fc2_w16.shape[0] == 384) else None
        bp.num_heads = self.attn.num_heads
        bp.head_gate, bp.neuron_gate = self.attn.gate_on(device), self.mlp.gate_on(device)
        bp.dp_prob = self.drop_prob if self.training else 0.
        bp.module = self
        bp.masters = bp.finish = None
        if bp.qkv_b is None:
            raise NotImplementedError("qkv_bias=False is not used by any registered DeViT model")
        return bp

    def forward(self, x, output_qkv=False, output_att=False):
        x_out, qkvs, atts, _ = run_blocks([self], x, self.training, output_qkv, output_att, False)
        outputs = {'output': x_out}
        outputs['qkv'] = qkvs[0] if output_qkv else None
        outputs['attention'] = atts[0] if output_att else None
        return outputs


class DropPath(nn.Module):
    """timm DropPath semantics (= models/utils/stochastic_depth.py:8-25): per-sample floor(keep + U)/keep."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        raise RuntimeError("DropPath is folded into the residual GEMM epilogue; it is never called directly")


_keep_cache = {}


def draw_dp_scales(blocks_params, B, device, training):
    """Per-sample stochastic-depth scales floor(keep + u) / keep (models/de_vit.py:114-121, timm DropPath) for both
    branches of every block, drawn with ONE rand call per forward."""
    probs = [bp.dp_prob if training else 0. for bp in blocks_params]
    if not any(p > 0. for p in probs):
        return None
    key = (tuple(probs), str(device))
    keep = _keep_cache.get(key)
    if keep is None:
        keep = torch.tensor([1.0 - p for p in probs], dtype=torch.float32, device=device).view(-1, 1, 1)
        _keep_cache[key] = keep
    u = torch.rand((len(probs), 2, B), dtype=torch.float32, device=device)
    sc = torch.floor(keep + u) / keep
    return [(sc[i, 0], sc[i, 1]) if p > 0. else None for i, p in enumerate(probs)]


LEAN_TAIL = os.environ.get("DEVIT_LEAN_TAIL", "1") == "1"      # 0: lean_tail() is a no-op (A/B runs)


@contextlib.contextmanager
def lean_tail(*models):
    """Forwards of `models` inside this context promise that the caller reads nothing of the LAST block but the class /
    distillation tokens of its output (through 'output' / 'last_tokens'): not its q/k/v (the 'qkv' entry of the last block
    is None; DEKD reads the middle block's, engine.py:91-92), not `head_output` / `neuron_output` of its modules.  The
    model then runs that block on the token rows only (ops._tail_forward): the reference computes all 198 rows and
    keeps two (models/de_vit.py:286-288).  Logits are bit-identical.  engine.distill_forward / evaluate enter it."""
    vits = [m.module if hasattr(m, "module") else m for m in models]
    prev = [getattr(v, "_lean_tail", False) for v in vits]
    for v in vits:
        v._lean_tail = LEAN_TAIL
    try:
        yield
    finally:
        for v, p in zip(vits, prev):
            v._lean_tail = p


def run_blocks(blocks, x, training, want_qkv, want_att, want_enc, grad_ready=None, dp_scales="draw", exact_gelu=0,
               precision="bf16", qkv_pad_layers=None, lean_tokens=0):
    """Run a list of Blocks as one EncoderFn node.  Returns (x, qkv tuples, att tensors, enc tensors).
    lean_tokens > 0 (see lean_tail): x comes back as [B, lean_tokens, D] and the last block's qkv entry is None."""
    L.require_device(x)
    if x.dtype != torch.float32:
        x = x.float()
    B, N, D = x.shape
    if precision == "f16" and torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for b in blocks for p in b.parameters())):
        raise L.DevitError('precision="f16" is the frozen-teacher forward: no backward kernels read f16 activations; run it '
                           'under torch.no_grad() with requires_grad off, or use precision="bf16"')
    if any(getattr(b, "_compact", None) is not None and b._compact.get("trainable") for b in blocks):
        from . import shrink             # the optimizer rewrote the masters (and their bf16 copies) since the last forward
        shrink.refresh_blocks(blocks)
    bps = [b.block_params(x.device, precision == "f16") for b in blocks]
    if dp_scales == "draw":
        dp_scales = draw_dp_scales(bps, B, x.device, training)
    nb = len(blocks)
    lean = lean_tokens if (lean_tokens and nb >= 2 and precision != "f32" and not want_att and not want_enc and
                           (not want_qkv or (qkv_pad_layers is not None and nb - 1 not in qkv_pad_layers))) else 0
    cfg = ops.EncoderCfg(bps, training, dp_scales, want_qkv, want_att, want_enc, exact_gelu=exact_gelu,
                         grad_ready=grad_ready, qkv_pad_layers=qkv_pad_layers, lean_tokens=lean)
    cfg.grad_enabled = torch.is_grad_enabled()
    flat = [p for bp in bps for p in bp.all_params()]
    if precision == "f32":
        from . import ops_f32
        if any(bp.compacted for bp in bps):
            raise L.DevitError('precision="f32" runs the uncompacted weights: call shrink.uncompact(model) first')
        outs = ops_f32.EncoderF32Fn.apply(x, cfg, *flat)
    else:
        outs = ops.EncoderFn.apply(x, cfg, *flat)
    i = 1
    qkvs = atts = encs = None
    if want_qkv:
        nq = nb - 1 if lean else nb
        qkvs = [qkv_views(t, B, N, bps[j].num_heads) for j, t in enumerate(outs[i:i + nq])] + [None] * (nb - nq)
        i += nq
    if want_att:
        atts = [t.view(B, N, D) for t in outs[i:i + nb]]
        i += nb
    if want_enc:
        encs = list(outs[i:i + nb])
    return outs[0], qkvs, atts, encs


def _standalone_mlp(m, x):
    """Mlp.forward as its own autograd node (module-level API / tests; the model path uses EncoderFn)."""
    return ops.MlpFn.apply(x, m, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, _w16(m.fc1), _w16(m.fc2),
                           m.gate_on(x.device) if x.is_cuda else None, torch.is_grad_enabled())


def _standalone_attention(m, x, output_qkv):
    out, qkv = ops.AttentionFn.apply(x, m, m.qkv.weight, m.qkv.bias, m.proj.weight, m.proj.bias, _w16(m.qkv),
                                     _w16(m.proj), m.gate_on(x.device) if x.is_cuda else None,
                                     torch.is_grad_enabled())
    B, N, _ = x.shape
    outputs = {'output': out}
    outputs['qkv'] = qkv_views(qkv, B, N, m.num_heads) if output_qkv else None
    return outputs


class PatchEmbed(nn.Module):
    """timm 0.5.4 PatchEmbed surface: Conv2d(3, D, 16, 16) weights, run as im2row + MFMA GEMM."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()
        if img_size != 224 or patch_size != 16 or in_chans != 3:
            raise NotImplementedError("the patch-embed kernels are built for 3x224x224 images, 16x16 patches")


class VisionTransformer(nn.Module):
    """models/de_vit.py:124-334 (gated ViT / DeiT with dict outputs)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, representation_size=None, distilled=False,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., embed_layer=PatchEmbed, norm_layer=None,
                 act_layer=None, weight_init='', resize_dim=None):
        super().__init__()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 2 if distilled else 1
        self.resize_dim = resize_dim
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        act_layer = act_layer or nn.GELU
        if representation_size:
            raise NotImplementedError("representation_size (pre_logits) is not used by any DeViT model")

        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if distilled else None
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        if drop_rate != 0.:
            raise NotImplementedError("dropout p > 0 is not on the DeViT hot path (distill_sub.py --drop 0.0)")

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]      # de_vit.py:175
        self.blocks = nn.Sequential(*[
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, act_layer=act_layer)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.head_dist = None
        if distilled:
            self.head_dist = nn.Linear(self.embed_dim, self.num_classes) if num_classes > 0 else nn.Identity()
        if self.resize_dim is not None:   # registered for checkpoint compatibility (de_vit.py:198-201)
            self.resize_mlp = nn.Linear(self.embed_dim, self.resize_dim)
            self.resize_att_mlp = nn.Linear(self.embed_dim, self.resize_dim)
            self.resize_encoder_mlp = nn.Linear(self.embed_dim, self.resize_dim)
        self.grad_ready = None      # set by devit_amd.ddp.BucketedGradReducer
        self.exact_gelu = 0
        # "bf16": the training path.  "f16": the same kernels with IEEE f16 operands / stored activations, forward only --
        # for frozen teachers (DeiT-B logits 1.1e-3 from fp32 instead of 6.8e-3; step 1.4 % slower: both measured).  "f32": exact-fp32 parity
        # path (ops_f32.py), not tuned
        self._pinned_precision = None
        self.precision = "bf16"
        self.init_weights(weight_init)

    @property
    def precision(self):
        return self._precision

    @precision.setter
    def precision(self, value):
        if value not in ("bf16", "f16", "f32"):
            raise ValueError(f"precision {value!r}: one of 'bf16', 'f16', 'f32'")
        pinned = getattr(self, "_pinned_precision", None)
        if pinned is not None and value != pinned:
            raise L.DevitError(f"this geometry (embed_dim {self.embed_dim}) runs on the exact-fp32 kernels only: precision is pinned to "
                               f"{pinned!r} (de_vit.check_geometry)")
        self._precision = value

    def request_precision(self, value):
        """What a CLI does with its --teacher-precision flag: set it where the model has a choice; a model whose kernel family is pinned
        (the D = 192 names run on the exact-fp32 kernels only) keeps it and says so once -- `--teacher-model deit_tiny_*` must not die on a
        flag whose only values are 16-bit types (advisor r05)."""
        pinned = getattr(self, "_pinned_precision", None)
        if pinned is not None and value != pinned:
            import warnings
            warnings.warn(f"{type(self).__name__} (embed_dim {self.embed_dim}) runs on the {pinned!r} kernels only: --teacher-precision {value} ignored")
            return pinned
        self.precision = value
        return value

    def pin_precision(self, value):
        """Fix the kernel family this model runs on (narrow geometries: "f32")."""
        self._pinned_precision = None
        self.precision = value
        self._pinned_precision = value

    # ---- init / bookkeeping identical to the reference (de_vit.py:205-240) ------------------------------
    def init_weights(self, mode=''):
        assert mode in ('jax', 'jax_nlhb', 'nlhb', '')
        if mode.startswith('jax'):
            raise NotImplementedError("jax weight init is not used by DeViT")
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        if self.dist_token is not None:
            nn.init.trunc_normal_(self.dist_token, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.apply(_init_vit_weights)

    def _init_weights(self, m):
        _init_vit_weights(m)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'dist_token'}

    def get_classifier(self):
        return self.head if self.dist_token is None else (self.head, self.head_dist)

    def reset_classifier(self, num_classes, global_pool=''):
        self.num_classes = num_classes
        dev = self.cls_token.device
        self.head = (nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()).to(dev)
        if self.num_tokens == 2:
            self.head_dist = (nn.Linear(self.embed_dim, self.num_classes) if num_classes > 0 else nn.Identity()).to(dev)

    # ---- forward --------------------------------------------------------------------------------------
    def embed(self, x):
        """patch_embed + cls/dist tokens + pos_embed (de_vit.py:258-264) -> fp32 [B, T, D]."""
        L.require_device(x.rows if isinstance(x, ops.PatchRows) else x)
        if self.precision == "f32":
            from . import ops_f32
            if isinstance(x, ops.PatchRows):
                raise L.DevitError('precision="f32" reads the fp32 images: pass the image tensor, not bf16 PatchRows')
            return ops_f32.PatchEmbedF32Fn.apply(x, self.patch_embed.proj.weight, self.patch_embed.proj.bias,
                                                 self.cls_token, self.dist_token, self.pos_embed, self.grad_ready)
        return ops.PatchEmbedFn.apply(x, self.patch_embed.proj.weight, self.patch_embed.proj.bias, self.cls_token,
                                      self.dist_token, self.pos_embed, _w16(self.patch_embed.proj, self.precision == "f16"),
                                      self.grad_ready)

    def _tokens_and_logits(self, x, with_heads):
        head = self.head if (with_heads and isinstance(self.head, nn.Linear)) else None
        hd = self.head_dist if (with_heads and isinstance(self.head_dist, nn.Linear)) else None
        return ops.HeadsFn.apply(x, self.norm.weight, self.norm.bias,
                                 head.weight if head is not None else None, head.bias if head is not None else None,
                                 hd.weight if hd is not None else None, hd.bias if hd is not None else None,
                                 self.num_tokens, self.norm.eps, self.grad_ready)

    def forward_features(self, x, output_qkv=False, output_att=False, output_emb=False, output_encoders=False):
        out, _ = self._features(x, output_qkv, output_att, output_emb, output_encoders, with_heads=False)
        return out

    def _features(self, x, output_qkv, output_att, output_emb, output_encoders, with_heads):
        if self.resize_dim is not None:
            raise NotImplementedError("resize_dim (--distillation-token) is not built: the reference's own model cannot run it "
                                      "(models/de_vit.py:276 applies resize_att_mlp to None unless output_att, :313-314 to the distilled "
                                      "(cls, dist) tuple; train_subdata.py:253 unpacks the returned dict as a pair) -- DESIGN.md section 9")
        x = self.embed(x)
        emb = x
        xo, qkvs, atts, encs = run_blocks(list(self.blocks), x, self.training, output_qkv, output_att, output_encoders,
                                          grad_ready=self.grad_ready, exact_gelu=self.exact_gelu,
                                          precision=self.precision,
                                          qkv_pad_layers=getattr(self, "qkv_pad_layers", None),
                                          lean_tokens=self.num_tokens if getattr(self, "_lean_tail", False) else 0)
        depth = len(self.blocks)
        encoder_outputs = [emb] if output_emb else []
        encoder_outputs += encs if output_encoders else [None] * depth
        heads = self._tokens_and_logits(xo, with_heads)
        tok = heads[0]
        outputs = {'output': tok[:, 0] if self.dist_token is None else (tok[:, 0], tok[:, 1]),
                   'qkv': qkvs if output_qkv else [None] * depth,
                   'attention': atts if output_att else [None] * depth,
                   'encoder': encoder_outputs}
        return outputs, heads

    def forward(self, x, distill_token=False, output_qkv=False, output_att=False, output_emb=False,
                output_encoders=False):
        outputs, heads = self._features(x, output_qkv, output_att, output_emb, output_encoders, with_heads=True)
        last_tokens = outputs['output']
        any_flag = distill_token or output_qkv or output_att or output_emb or output_encoders
        if self.head_dist is not None:
            x, x_dist = (heads[1], heads[2]) if len(heads) == 3 else last_tokens     # Identity heads: tokens
            outputs['output'] = (x, x_dist) if self.training else (x + x_dist) / 2     # de_vit.py:318
            outputs['last_tokens'] = last_tokens if distill_token else None
            if any_flag:
                return outputs
            return (x, x_dist) if self.training else outputs['output']
        x = heads[1] if len(heads) >= 2 else last_tokens
        outputs['output'] = x
        outputs['last_tokens'] = last_tokens if distill_token else None
        return outputs if any_flag else x


def _init_vit_weights(module, name='', head_bias=0., jax_impl=False):
    """models/de_vit.py:337-369 as reached from init_weights('') (no names -> heads are NOT zeroed)."""
    if isinstance(module, nn.Linear):
        nn.init.trunc_normal_(module.weight, std=.02)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif isinstance(module, (nn.LayerNorm, nn.GroupNorm, nn.BatchNorm2d)):
        nn.init.zeros_(module.bias)
        nn.init.ones_(module.weight)


# Geometry table, importable and correct (the reference's models/utils/config.py:1-17 raises NameError and
# lists 192/3 for `dedeit`; App. D Q1).  `distilled` follows models/deit_vit.py:528-550.
_LN = partial(nn.LayerNorm, eps=1e-6)
model_config = {
    'dedeit': dict(patch_size=16, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True, norm_layer=_LN, distilled=True),
    'devit': dict(patch_size=16, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True, norm_layer=_LN, distilled=False),
    'deit_tiny_patch16_224': dict(patch_size=16, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, qkv_bias=True, norm_layer=_LN, distilled=False),
    'deit_base_patch16_224': dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, norm_layer=_LN, distilled=False),
    'deit_tiny_distilled_patch16_224': dict(patch_size=16, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, qkv_bias=True, norm_layer=_LN, distilled=True),
    'deit_base_distilled_patch16_224': dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, norm_layer=_LN, distilled=True),
    'vit_tiny_patch16_224': dict(patch_size=16, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, qkv_bias=True, norm_layer=_LN, distilled=False),
    'vit_base_patch16_224': dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, norm_layer=_LN, distilled=False),
    'vit_large_patch16_224': dict(patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True, norm_layer=_LN, distilled=False),
}


def check_geometry(name, embed_dim, num_heads, mlp_ratio=4.):
    """Refuse, at construction and with the reason, a geometry the gfx950 kernels are not built for -- instead of registering a
    name whose first forward fails deep inside a launch.  Heads are 64 wide everywhere (csrc/attention.hip and the reference's own
    registered names).  The MFMA GEMM tiles, the fused attention kernels and the block calls want every width in multiples of 128
    (csrc/gemm.hip, encoder.hip): geometries that are multiples of 64 only -- the three D = 192 names of models/deit_vit.py:457-525
    (`deit_tiny_patch16_224`, `deit_tiny_distilled_patch16_224`, `vit_tiny_patch16_224`; none of them is on the DeViT path: student
    `dedeit` is 384 / 6, teachers are DeiT-B 768 / 12 or ViT-L 1024 / 16, README.md:50-68 of the reference) -- run on the exact-fp32
    kernels instead (csrc/sgemm.hip, ops_f32.py: forward, backward, every loss; a few TFLOP/s, which a 1.3-GFLOP model does not notice):
    their `precision` is pinned to "f32".  Returns True for such a narrow geometry."""
    hidden = int(embed_dim * mlp_ratio)
    if embed_dim % 64 or embed_dim != num_heads * 64 or hidden % 64 or embed_dim > 1024:
        raise NotImplementedError(
            f"{name}: embed_dim={embed_dim}, num_heads={num_heads}, hidden={hidden} is not a geometry the MI355X kernels are built "
            f"for (embed_dim and hidden must be multiples of 64, heads 64 wide, embed_dim <= 1024); registered names: {list(model_config)}")
    return bool(embed_dim % 128 or hidden % 128)


def _cfg(**kwargs):
    return {'url': '', 'num_classes': 1000, 'input_size': (3, 224, 224), 'pool_size': None, 'crop_pct': .9,
            'interpolation': 'bicubic', 'fixed_input_size': True, 'mean': (0.485, 0.456, 0.406),
            'std': (0.229, 0.224, 0.225), 'first_conv': 'patch_embed.proj', 'classifier': 'head', **kwargs}


def _make(name):
    def fn(pretrained=False, pretrained_path=None, **kwargs):
        geo = {**model_config[name], **kwargs}
        narrow = check_geometry(name, geo.get('embed_dim', 768), geo.get('num_heads', 12), geo.get('mlp_ratio', 4.))
        model = VisionTransformer(**geo)
        if narrow:
            model.pin_precision("f32")
        model.default_cfg = _cfg()
        if pretrained_path is not None and pretrained:
            ckpt = torch.load(pretrained_path, map_location='cpu', weights_only=False)
            model.load_state_dict(ckpt['model'] if 'model' in ckpt else ckpt)
        return model
    fn.__name__ = name
    fn.__doc__ = f"{name}: registered like models/de_vit.py:495-513 / models/deit_vit.py:457-525 (dict-API class)."
    return register_model(fn)


dedeit = _make('dedeit')
devit = _make('devit')
for _n in list(model_config):
    if _n not in ('dedeit', 'devit'):
        globals()[_n] = _make(_n)
