#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules on CPU fp32.

Runs only in the build container (needs /root/reference; never on the GPU box).
Nothing of the reference is copied: the script imports it in place, feeds it the
deterministic inputs/weights of oracle/detgen.py + oracle.devit_oracle.make_state
and stores OUTPUTS only.

Harness-side shims (no edit of /root/reference):
  1. builtins.partial / builtins.nn  -- models/utils/config.py:4 uses them unimported.
  2. stand-in `timm` modules -- timm==0.5.4 (README.md:18) is not installed and not
     vendored.  Only PatchEmbed (Conv2d 16/16 + flatten/transpose), DropPath (delegates
     to the reference's in-tree models/utils/stochastic_depth.py:8-25) and
     SoftTargetCrossEntropy carry arithmetic.
  3. torch.Tensor.get_device -> self.device for CPU tensors (models/de_vit.py:42,78).

Usage:  python tests/golden/make_golden.py            (writes next to this file)
"""
import builtins
import functools
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle.detgen import det_array, det_labels  # noqa: E402
from oracle import devit_oracle as O  # noqa: E402


# ----------------------------------------------------------------------------- shims
def install_shims():
    builtins.partial = functools.partial
    builtins.nn = nn

    spec = importlib.util.spec_from_file_location("_ref_sd", f"{REF}/models/utils/stochastic_depth.py")
    sd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sd)

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
            super().__init__()
            self.img_size = (img_size, img_size)
            self.patch_size = (patch_size, patch_size)
            self.grid_size = (img_size // patch_size, img_size // patch_size)
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.flatten = flatten
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
            self.norm = nn.Identity()

        def forward(self, x):
            x = self.proj(x)
            x = x.flatten(2).transpose(1, 2)
            return self.norm(x)

    class SoftTargetCrossEntropy(nn.Module):
        def forward(self, x, target):
            return torch.sum(-target * torch.nn.functional.log_softmax(x, dim=-1), dim=-1).mean()

    registry = {}

    def register_model(fn):
        registry[fn.__name__] = fn
        return fn

    def create_model(model_name, pretrained=False, **kw):
        kw = {k: v for k, v in kw.items() if v is not None}
        return registry[model_name](pretrained=pretrained, **kw)

    def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
        return nn.init.trunc_normal_(t, mean, std, a, b)

    def named_apply(fn, module, name="", depth_first=True, include_root=False):
        for cn, cm in module.named_children():
            named_apply(fn, cm, ".".join((name, cn)) if name else cn, depth_first, True)
        if include_root:
            fn(module=module, name=name)
        return module

    def mk(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mk("timm")
    mk("timm.models", create_model=create_model)
    mk("timm.models.registry", register_model=register_model)
    mk("timm.models.vision_transformer", _cfg=lambda **kw: dict(kw))
    mk("timm.models.layers", PatchEmbed=PatchEmbed, DropPath=sd.DropPath, trunc_normal_=trunc_normal_,
       lecun_normal_=trunc_normal_, Mlp=None)
    mk("timm.models.helpers", named_apply=named_apply, adapt_input_conv=None)
    mk("timm.loss", SoftTargetCrossEntropy=SoftTargetCrossEntropy)

    orig = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: self.device if self.device.type == "cpu" else orig(self)
    sys.path.insert(0, REF)
    return registry, create_model, sd


class RandQueue:
    """Replaces torch.rand inside DropPath so the masks are data we control."""

    def __init__(self, name, keep_probs):
        self.name, self.n, self.scales, self.keep = name, 0, [], keep_probs

    def __call__(self, shape, dtype=None, device=None):
        u = det_array(f"{self.name}/{self.n}", tuple(shape), std=1.0)
        u = torch.from_numpy(np.abs(u) % 1.0).to(dtype or torch.float32)
        kp = self.keep[self.n]
        self.scales.append(torch.floor(kp + u).reshape(-1) / kp)
        self.n += 1
        return u


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def load_into(model, st):
    sd = model.state_dict()
    assert list(sd.keys()) == list(st.keys()), "state_dict key order differs from oracle.state_keys"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(st[k].shape), k
    model.load_state_dict(st)


def ens_state(tag="ENS"):
    """Deterministic EnsMLP weights in the reference's registration order (models/ensemble_models.py:55-63)."""
    shapes = [("cls_mlp.weight", (768, 1536)), ("cls_mlp.bias", (768,)), ("cls_classifier.weight", (100, 768)),
              ("cls_classifier.bias", (100,)), ("dist_mlp.weight", (768, 1536)), ("dist_mlp.bias", (768,)),
              ("dist_classifier.weight", (100, 768)), ("dist_classifier.bias", (100,))]
    return {k: torch.from_numpy(det_array(f"{tag}/{k}", sh, std=0.02)) for k, sh in shapes}


def make_ensemble(create_model, de_vit, ref_losses):
    """ensemble stage (config 5): MultiViT(4 x dedeit) + EnsMLP + EnsLoss on bs 4, C = 100."""
    import models.ensemble_models as em          # reference
    gs, gt = O.GEOMETRY["dedeit"], O.GEOMETRY["deit_base_distilled_patch16_224"]
    multi = em.MultiViT(model="dedeit", drop=0, drop_path=0.0, num_classes_list=[25] * 4, num_div=4)
    ens = em.EnsMLP(model="dedeit", num_class=100, sub_size=384, num_classes_list=[25] * 4, teacher_size=768)
    sd = multi.state_dict()
    keys = list(sd.keys())
    for i in range(4):                            # positional copy like ensemble.py:192-200,229-238
        sub = O.make_state(gs, 25, f"E{i}")
        src = list(sub.keys())
        for j in range(len(src) - 4):
            sd[keys[i * (len(src) - 4) + j]] = sub[src[j]]
    multi.load_state_dict(sd)
    ens.load_state_dict(ens_state())
    teacher = de_vit.VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                                       norm_layer=functools.partial(nn.LayerNorm, eps=1e-6), distilled=True, num_classes=100)
    teacher.load_state_dict(O.make_state(gt, 100, "T100"))
    teacher.eval()
    img = torch.from_numpy(det_array("img4", (4, 3, 224, 224)))
    multi.eval(); ens.eval()
    with torch.no_grad():
        logits_eval = ens(multi(img))
    multi.train(); ens.train()
    feats = multi(img)
    tokens, logits = ens(feats, True)
    y = det_labels("ens_y", 4, 100)
    soft = torch.full((4, 100), 0.1 / 100).scatter_(1, torch.from_numpy(y)[:, None], 0.9 + 0.1 / 100)
    crit = ref_losses.EnsLoss(sys.modules["timm.loss"].SoftTargetCrossEntropy(), teacher, "dedeit", "hard", 0.5, 1.0)
    token_loss, cls_loss = crit(img, (tokens, logits), soft)
    (token_loss + cls_loss).backward()
    save("ensemble", logits_eval=logits_eval, logits_train=logits, tok_cls=tokens[0], tok_dist=tokens[1],
         token_loss=token_loss, cls_loss=cls_loss, soft_targets=soft,
         g_cls_mlp_w_rows=ens.cls_mlp.weight.grad[::48], g_dist_cls_w=ens.dist_classifier.weight.grad[::10],
         g_b2_fc1_rows=multi.backbones[2].blocks[3].mlp.fc1.weight.grad[::96],
         g_b0_pos=multi.backbones[0].pos_embed.grad[0, ::16], n_multi_keys=np.array(len(keys)))



def make_misc(ref_losses):
    """Round-2 fixtures: FLOP / parameter known answers (core/compute_metric.py), DeiT DistillationLoss
    (utils/losses.py:44-119) with every base criterion train_subdata.py:409-416 can pick, the importance ranking of
    core/imp_rank.py:16-47,93-129 on small synthetic activations, and Mixup / CutMix images + targets from timm's
    formulas (timm is not installed: lambda and the box are inputs, so no RNG parity is involved)."""
    spec = importlib.util.spec_from_file_location("_ref_metric", f"{REF}/core/compute_metric.py")
    cm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cm)
    z = [0.0] * 12
    s = dict(emb=384, seq_length=197, mlp_ratio=4, head=6, layer=12, num_class=1000)
    flops = {
        "dedeit_dense_gflops": cm.cal_shrink_flops(z, z, **s), "dedeit_dense_mparams": cm.cal_shrink_paras(z, z, **s),
        "deitb_dense_gflops": cm.cal_shrink_flops(z, z), "deitb_dense_mparams": cm.cal_shrink_paras(z, z),
        "dedeit_shrunk_0.3_gflops": cm.cal_shrink_flops([0.3] * 12, [0.3] * 12, **s),
        "dedeit_shrunk_0.3_mparams": cm.cal_shrink_paras([0.3] * 12, [0.3] * 12, **s),
        "dedeit_c25_n198_gflops": cm.cal_shrink_flops(z, z, **{**s, "seq_length": 198, "num_class": 25}),
        "deitb_c25_n198_gflops": cm.cal_shrink_flops(z, z, seq_length=198, num_class=25),
        "dedeit_mixed_gflops": cm.cal_shrink_flops([0.1 * (i % 4) for i in range(12)], [0.17 * (i % 3) for i in range(12)], **s),
    }
    with open(os.path.join(HERE, "flops.json"), "w") as f:
        json.dump(flops, f, indent=0)
    print("flops.json", flops)

    # ---- repeated-augmentation sampler (utils/samplers.py:8-63): the index stream per (dataset length, world, rank, epoch)
    spec = importlib.util.spec_from_file_location("_ref_samplers", f"{REF}/utils/samplers.py")
    sm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sm)

    class _Len:
        def __init__(self, n): self.n = n
        def __len__(self): return self.n
    cases = []
    for n, world in ((1000, 1), (1000, 4), (2570, 8), (513, 2), (256, 1), (255, 1)):
        for rank in sorted({0, world - 1}):
            for epoch in (0, 3):
                smp = sm.RASampler(_Len(n), num_replicas=world, rank=rank, shuffle=True)
                smp.set_epoch(epoch)
                idx = list(smp)
                cases.append(dict(n=n, world=world, rank=rank, epoch=epoch, length=len(smp), head=idx[:24],
                                  checksum=int(sum((i + 1) * (v + 1) for i, v in enumerate(idx)) % 1000000007)))
    with open(os.path.join(HERE, "ra_sampler.json"), "w") as f:
        json.dump(cases, f)
    print("ra_sampler.json", len(cases), "cases")

    # ---- DistillationLoss (teacher inside the criterion) ------------------------------------------
    B, C = 8, 25
    lo = torch.from_numpy(det_array("dl/lo", (B, C), std=1.5)).requires_grad_(True)
    lk = torch.from_numpy(det_array("dl/lk", (B, C), std=1.5)).requires_grad_(True)
    lt = torch.from_numpy(det_array("dl/lt", (B, C), std=2.0))
    y = torch.from_numpy(det_labels("dl/y", B, C))
    soft = torch.softmax(torch.from_numpy(det_array("dl/soft", (B, C), std=2.0)), 1)

    class Teacher(nn.Module):
        def forward(self, x, *a):
            return lt
    out = dict(lo=lo, lk=lk, lt=lt, y=y, soft=soft)
    bases = {"ce": (nn.CrossEntropyLoss(), y), "ls": (ref_losses.LabelSmoothingCrossEntropy(0.1), y),
             "soft": (sys.modules["timm.loss"].SoftTargetCrossEntropy(), soft)}
    for bname, (base, labels) in bases.items():
        for kind, tau in (("none", 1.0), ("hard", 1.0), ("soft", 3.0)):
            crit = ref_losses.DistillationLoss(base, Teacher(), kind, 0.5, tau, False)
            loss = crit(torch.zeros(B, 3, 8, 8), (lo, lk), labels)
            g = torch.autograd.grad(loss, [lo, lk], allow_unused=True)
            out[f"{bname}_{kind}_loss"] = loss
            out[f"{bname}_{kind}_dlo"] = g[0]
            out[f"{bname}_{kind}_dlk"] = g[1] if g[1] is not None else torch.zeros_like(lk)
    save("distillation_loss", **out)

    # ---- importance ranking (core/imp_rank.py) on synthetic activations --------------------------
    spec = importlib.util.spec_from_file_location("_ref_rank", f"{REF}/core/imp_rank.py")
    ir = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ir)
    nn.Module.cuda = lambda self, *a, **k: self          # the reference hard-codes .cuda() (imp_rank.py:17-18,26)
    torch.Tensor.cuda = lambda self, *a, **k: self
    Bn, Nn, Hid, H, hd, Cn = 6, 20, 48, 4, 8, 10

    class Mlp(nn.Module):                                 # discovered by `'Mlp' in str(m)` (imp_rank.py:30)
        def __init__(self, i):
            super().__init__()
            self.neuron_output = torch.from_numpy(det_array(f"rank/n{i}", (Bn, Nn, Hid), std=0.8))
            self.hidden_features = Hid

    class Attention(nn.Module):
        def __init__(self, i):
            super().__init__()
            self.head_output = torch.from_numpy(det_array(f"rank/h{i}", (Bn, Nn, H, hd), std=0.8))
            self.num_heads = H

    class Fake(nn.Module):
        def __init__(self):
            super().__init__()
            self.a0, self.m0, self.a1, self.m1 = Attention(0), Mlp(0), Attention(1), Mlp(1)
            self.logits = torch.from_numpy(det_array("rank/logits", (Bn, Cn), std=1.5))

        def forward(self, x):
            return self.logits
    fake = Fake()
    loader = [(torch.zeros(Bn, 3, 4, 4), torch.zeros(Bn, dtype=torch.long))]
    nrank = ir.mlp_neuron_rank(fake, loader)
    hrank = ir.attn_head_rank(fake, loader)
    nmask = ir.mlp_neuron_mask(fake, [0.3, 0.5], nrank)
    hmask = ir.attn_head_mask(fake, [0.3, 0.5], hrank)
    save("imp_rank", logits=fake.logits, n0=fake.m0.neuron_output, n1=fake.m1.neuron_output, h0=fake.a0.head_output,
         h1=fake.a1.head_output, neuron_rank=np.stack(nrank), head_rank=np.stack(hrank),
         neuron_mask=torch.stack(nmask), head_mask=torch.stack(hmask), sparsity=np.array([0.3, 0.5]))

    # ---- Mixup / CutMix (timm.data.Mixup, mode='batch') ---------------------------
    # RESTATED, NOT PINNED: timm is not installed here, so these vectors come from the formulas below, not from timm's code.
    Bm, Cm, eps = 4, 10, 0.1
    img = det_array("mix/img", (Bm, 3, 224, 224))
    yy = det_labels("mix/y", Bm, Cm)
    off, on = eps / Cm, 1 - eps + eps / Cm
    oh = np.full((Bm, Cm), off, np.float32)
    oh[np.arange(Bm), yy] = on
    lam = 0.37
    mix_img = img * lam + img[::-1] * (1 - lam)
    mix_t = oh * lam + oh[::-1] * (1 - lam)
    y0, y1, x0, x1 = 30, 141, 64, 200
    cut_img = img.copy()
    cut_img[:, :, y0:y1, x0:x1] = img[::-1][:, :, y0:y1, x0:x1]
    lam_c = 1.0 - (y1 - y0) * (x1 - x0) / float(224 * 224)
    cut_t = oh * lam_c + oh[::-1] * (1 - lam_c)
    sub = lambda a: a[:, :, ::7, ::5]                     # keeps the fixture small; the box edges fall between samples
    save("mixup", y=yy, lam=np.float32(lam), box=np.array([y0, y1, x0, x1]), lam_cut=np.float32(lam_c), smoothing=np.float32(eps),
         mix_img=sub(mix_img), mix_targets=mix_t, cut_img=sub(cut_img), cut_targets=cut_t,
         cut_rows=cut_img[:, :, 28:32, 60:68], cut_rows2=cut_img[:, :, 139:143, 196:204])


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    _, create_model, _ = install_shims()
    import models.de_vit as de_vit          # noqa: E402  (reference)
    import utils.losses as ref_losses       # noqa: E402  (reference)
    if "--only-ensemble" in sys.argv:
        make_ensemble(create_model, de_vit, ref_losses)
        return
    if "--only-misc" in sys.argv:
        make_misc(ref_losses)
        return

    C = 25
    gs, gt = O.GEOMETRY["dedeit"], O.GEOMETRY["deit_base_distilled_patch16_224"]
    st_s, st_t = O.make_state(gs, C, "S"), O.make_state(gt, C, "T")

    student = create_model("dedeit", pretrained=False, num_classes=C, drop_rate=0.0, drop_path_rate=0.0,
                           drop_block_rate=None)
    # teacher = the dict-API class with DeiT-B hyper-parameters
    teacher = de_vit.VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                                       qkv_bias=True, norm_layer=functools.partial(nn.LayerNorm, eps=1e-6),
                                       distilled=True, num_classes=C)
    load_into(student, st_s)
    load_into(teacher, st_t)

    # ---- statedict / API contract ------------------------------------------------
    contract = {
        "dedeit_keys": [[k, list(v.shape)] for k, v in student.state_dict().items()],
        "deitb_keys": [[k, list(v.shape)] for k, v in teacher.state_dict().items()],
        "n_mlp": sum(1 for m in student.modules() if "Mlp" in str(m) and "Attention" not in str(m)),
        "n_attn": sum(1 for m in student.modules() if "Attention" in str(m) and "Mlp" not in str(m)),
        "n_params_dedeit_c25": sum(p.numel() for p in student.parameters()),
        "n_params_deitb_c25": sum(p.numel() for p in teacher.parameters()),
        "no_weight_decay": sorted(student.no_weight_decay()),
    }
    with open(os.path.join(HERE, "statedict_keys.json"), "w") as f:
        json.dump(contract, f, indent=0)

    img = torch.from_numpy(det_array("img8", (8, 3, 224, 224)))

    # ---- module fixtures (block 5 of each model, x ~ LayerNorm-scale input) -----------
    for tag, model, D, H in (("S", student, 384, 6), ("T", teacher, 768, 12)):
        x = torch.from_numpy(det_array(f"x/{tag}", (2, 198, D), std=1.0)).requires_grad_(True)
        blk = model.blocks[5]
        blk.eval()
        ng = torch.from_numpy((np.abs(det_array(f"ng/{tag}", (4 * D,))) > 0.4).astype(np.float32))
        hg = torch.ones(H)
        hg[1] = 0.0
        # Mlp, gate ones / masked
        y1 = blk.mlp(x)
        gx, gw = torch.autograd.grad(y1.square().sum(), [x, blk.mlp.fc1.weight])
        blk.mlp.gate = ng
        y2 = blk.mlp(x)
        nout = blk.mlp.neuron_output.detach().clone()
        blk.mlp.gate = torch.ones(4 * D)
        # Attention, gate ones / head 1 zeroed
        a1 = blk.attn(x, True)
        ga, = torch.autograd.grad(a1["output"].square().sum(), [x])
        blk.attn.gate = hg
        a2 = blk.attn(x, True)
        hout = blk.attn.head_output.detach().clone()
        blk.attn.gate = torch.ones(H)
        b1 = blk(x, output_qkv=True, output_att=True)
        sub = lambda t: t[:, ::9]   # 22 of the 198 tokens: keeps each fixture < 1 MB
        save(f"module_{tag}", mlp_y=sub(y1), mlp_dx=sub(gx), mlp_dw1_rows=gw[:8], mlp_y_gated=sub(y2),
             mlp_neuron_output_sum=nout.sum(dim=(0, 1)), neuron_gate=ng,
             attn_y=sub(a1["output"]), attn_q=a1["qkv"][0][:, :, :16], attn_k=a1["qkv"][1][:, :, :16],
             attn_v=a1["qkv"][2][:, :, :16], attn_dx=sub(ga), attn_y_gated=sub(a2["output"]),
             attn_head_output_sum=hout.sum(dim=(0, 1)), head_gate=hg,
             block_y=sub(b1["output"]), block_att=sub(b1["attention"]))

    # ---- whole-model fixtures ------------------------------------------------------
    for tag, model in (("dedeit", student), ("deitb", teacher)):
        model.eval()
        with torch.no_grad():
            logits = model(img)
            d = model(img, distill_token=True, output_qkv=True, output_att=True, output_emb=True,
                      output_encoders=True)
        assert torch.equal(d["output"], logits)
        q, k, v = d["qkv"][5]
        model.train()
        with torch.no_grad():
            tr = model(img)
        enc_stats = np.stack([[e.mean().item(), e.abs().mean().item()] for e in d["encoder"]])
        save(f"model_{tag}", logits=logits, top1=logits.argmax(1), train_cls=tr[0], train_dist=tr[1],
             q5=q[:2, :, :24], k5=k[:2, :, :24], v5=v[:2, :, :24], att5=d["attention"][5][:2, :24],
             enc_last=d["encoder"][-1][:2, :24], enc_stats=enc_stats,
             last_cls=d["last_tokens"][0], last_dist=d["last_tokens"][1])
        # bf16 autocast run of the same reference module: defines the bf16-mode tolerance
        model.eval()
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            lb = model(img)
        save(f"model_{tag}_bf16", logits=lb.float())

    # ---- loss fixtures -------------------------------------------------------------
    lo = torch.from_numpy(det_array("lo", (8, C), std=1.5)).requires_grad_(True)
    lk = torch.from_numpy(det_array("lk", (8, C), std=1.5)).requires_grad_(True)
    lt = torch.from_numpy(det_array("lt", (8, C), std=2.0))
    y1, y2 = det_labels("y1", 8, C), det_labels("y2", 8, C)
    eps, lam = 0.1, 0.7
    oh = lambda y: torch.full((8, C), eps / C).scatter_(1, torch.from_numpy(y)[:, None], 1 - eps + eps / C)
    soft = oh(y1) * lam + oh(y2) * (1 - lam)
    base = sys.modules["timm.loss"].SoftTargetCrossEntropy()
    res = {}
    for kind in ("hard", "soft"):
        crit = ref_losses.DistillLoss(base, kind, 0.5, 1.0 if kind == "hard" else 3.0)
        l = crit((lo, lk), lt, soft)
        g = torch.autograd.grad(l, [lo, lk])
        res.update({f"{kind}_loss": l, f"{kind}_dlo": g[0], f"{kind}_dlk": g[1]})
    save("loss_cls", soft_targets=soft, **res)
    # DistillLoss on hard (int64) labels with the two base criteria distill_sub.py:345-352 picks when mixup is off:
    # LabelSmoothingCrossEntropy(smoothing) (utils/losses.py:10-34) and nn.CrossEntropyLoss
    yl = torch.from_numpy(y1)
    res = {}
    for bname, base_h in (("ls", ref_losses.LabelSmoothingCrossEntropy(smoothing=0.1)), ("ce", nn.CrossEntropyLoss())):
        for kind in ("none", "hard", "soft"):
            crit = ref_losses.DistillLoss(base_h, kind, 0.5, 1.0 if kind != "soft" else 3.0)
            l = crit((lo, lk), lt, yl)
            g = torch.autograd.grad(l, [lo, lk], allow_unused=True)
            res.update({f"{bname}_{kind}_loss": l, f"{bname}_{kind}_dlo": g[0],
                        f"{bname}_{kind}_dlk": g[1] if g[1] is not None else torch.zeros_like(lk)})
    save("loss_cls_hardlabels", labels=yl, **res)

    tf = torch.from_numpy(det_array("tf", (2, 198, 3, 12, 64), std=0.25)).permute(2, 0, 3, 1, 4)[1]
    sf = torch.from_numpy(det_array("sf", (2, 198, 3, 6, 64), std=0.25)).permute(2, 0, 3, 1, 4)[1]
    sf = sf.detach().requires_grad_(True)
    l = ref_losses.feature_relation_loss(tf, sf)
    g, = torch.autograd.grad(l, [sf])
    save("loss_relation", loss=l, dstudent=g[:, :, ::9])

    # ---- one distillation step (engine.py:68-106) with recorded DropPath masks --------
    student_dp = create_model("dedeit", pretrained=False, num_classes=C, drop_rate=0.0, drop_path_rate=0.1,
                              drop_block_rate=None)
    load_into(student_dp, st_s)
    student_dp.train()
    teacher.eval()
    dpr = [x.item() for x in torch.linspace(0, 0.1, 12)]
    keep = []
    for i in range(1, 12):
        keep += [1 - dpr[i], 1 - dpr[i]]
    rq = RandQueue("dp", keep)
    y1, y2 = det_labels("sy1", 8, C), det_labels("sy2", 8, C)
    soft8 = oh(y1) * lam + oh(y2) * (1 - lam)
    crit = ref_losses.DistillLoss(base, "hard", 0.5, 1.0)
    real_rand = torch.rand
    torch.rand = rq
    try:
        outputs = student_dp(img, output_qkv=True)
    finally:
        torch.rand = real_rand
    with torch.no_grad():
        t_out = teacher(img, output_qkv=True)
    cls_loss = crit(outputs=outputs["output"], teacher_outputs=t_out["output"], labels=soft8)
    sq, tq = outputs["qkv"][12 // 2 - 1], t_out["qkv"][12 // 2 - 1]
    ql, kl, vl = [ref_losses.feature_relation_loss(tv, sv) / 12 for sv, tv in zip(sq, tq)]
    gama = (0.2, 0.1, 0.3)
    loss = cls_loss + gama[0] * ql + gama[1] * kl + gama[2] * vl
    loss.backward()
    names = [n for n, _ in student_dp.named_parameters()]
    gn = np.array([p.grad.norm().item() for _, p in student_dp.named_parameters()])
    grads = dict(student_dp.named_parameters())
    dp_scales = torch.stack([torch.ones(8), torch.ones(8)] + rq.scales).reshape(12, 2, 8)
    save("step_bs8", loss=loss, cls_loss=cls_loss, q_loss=ql, k_loss=kl, v_loss=vl, soft_targets=soft8,
         dp_scales=dp_scales, grad_norms=gn, teacher_logits=t_out["output"],
         stu_cls=outputs["output"][0], stu_dist=outputs["output"][1],
         g_head_w=grads["head.weight"].grad, g_head_dist_w=grads["head_dist.weight"].grad,
         g_qkv5_w_rows=grads["blocks.5.attn.qkv.weight"].grad[::48],
         g_fc1_0_rows=grads["blocks.0.mlp.fc1.weight"].grad[::64],
         g_fc2_11_rows=grads["blocks.11.mlp.fc2.weight"].grad[::16],
         g_pos=grads["pos_embed"].grad[0, ::8], g_cls=grads["cls_token"].grad,
         g_patch_w=grads["patch_embed.proj.weight"].grad[::16].reshape(-1, 768),
         g_norm_w=grads["norm.weight"].grad, g_ln1_5_b=grads["blocks.5.norm1.bias"].grad)
    with open(os.path.join(HERE, "step_param_names.json"), "w") as f:
        json.dump(names, f)
    make_ensemble(create_model, de_vit, ref_losses)
    make_misc(ref_losses)
    print("done")


if __name__ == "__main__":
    main()
