"""Model / loss / step parity on a real MI355X against (a) the golden vectors captured from the reference's
own modules (tests/golden/*.npz) and (b) the CPU oracle on fresh seeded inputs.

Tolerances.  The HIP path computes in bf16 (MFMA operands and stored branch activations; fp32 accumulate,
fp32 residual stream, fp32 LN / softmax / loss statistics).  The reference's own modules run under
torch.autocast(bfloat16) differ from their fp32 run by 1.2e-2 of max|logit| (tests/golden/model_*_bf16.npz,
SURVEY fact 8).  The bf16-mode bars are set from what the kernels measure on MI355X (profiles/*parity*): logits within 1.5e-2 of
max|logit| -- 1.9x the golden batch's 8.0e-3 / 7.6e-3 but only 1.6x the worst of seven other seeded batches (9.6e-3 for DeiT-B,
profiles/r04_e_parity_spread.json: the statistic is a maximum over 200 logits of bf16 rounding noise and moves from batch to batch) and
1.25x the reference's own bf16-autocast deviation --, top-1 indices bit-exact on the fixture set, the step's losses within
5e-4 relative (8e-5), gradient norms within 3e-3 (6e-4), gradient slices within 1e-2 of their max (3.8e-3); every check goes
through conftest.chk, which records value and bar (gpurun_out/parity_margins.json).  BASELINE.json's 1e-3 bar is asserted
on the exact-fp32 path (`precision="f32"`, test_f32_path_meets_1e3_bar; measured 2e-6)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import devit_oracle as O
from oracle.detgen import det_array
from conftest import chk

pytestmark = pytest.mark.gpu
C = 25
GS, GT = O.GEOMETRY["dedeit"], O.GEOMETRY["deit_base_distilled_patch16_224"]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


@pytest.fixture(scope="module")
def models(dev):
    import devit_amd
    st_s, st_t = O.make_state(GS, C, "S"), O.make_state(GT, C, "T")
    s = devit_amd.create_model("dedeit", num_classes=C, drop_path_rate=0.1, drop_block_rate=None)
    t = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=C)
    s.load_state_dict(st_s)
    t.load_state_dict(st_t)
    return s.to(dev), t.to(dev).eval(), st_s, st_t


def rel(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("which", ["dedeit", "deitb"])
def test_model_forward_vs_golden(golden, models, dev, which):
    s, t, _, _ = models
    m = s if which == "dedeit" else t
    g = golden(f"model_{which}")
    img = torch.from_numpy(det_array("img8", (8, 3, 224, 224))).to(dev)
    m.eval()
    with torch.no_grad():
        logits = m(img)
        d = m(img, distill_token=True, output_qkv=True, output_att=True, output_emb=True, output_encoders=True)
    assert isinstance(logits, torch.Tensor) and logits.shape == (8, C)
    e = rel(logits, g["logits"])
    assert chk(e, 1.5e-2), f"logits rel-to-max err {e:.3e}"        # measured 8.0e-3 / 7.6e-3 (profiles/r05_H_parity_margins.json; 9.6e-3 at worst over seven batches)
    assert np.array_equal(logits.argmax(1).cpu().numpy(), g["top1"])            # top-1 bit-exact
    assert set(d) == {"output", "qkv", "attention", "encoder", "last_tokens"}
    assert len(d["qkv"]) == 12 and len(d["attention"]) == 12 and len(d["encoder"]) == 13
    q, k, v = d["qkv"][5]
    D = m.embed_dim
    assert q.shape == (8, D // 64, 198, 64) and q.stride() == (198 * 3 * D, 64, 3 * D, 1)
    assert chk(rel(q[:2, :, :24], g["q5"]), 2e-2) and chk(rel(k[:2, :, :24], g["k5"]), 2e-2) and chk(rel(v[:2, :, :24], g["v5"]), 2e-2)
    assert chk(rel(d["attention"][5][:2, :24], g["att5"]), 2e-2)
    assert chk(rel(d["encoder"][-1][:2, :24], g["enc_last"]), 2e-2)
    assert chk(rel(d["last_tokens"][0], g["last_cls"]), 2e-2) and chk(rel(d["last_tokens"][1], g["last_dist"]), 2e-2)
    es = np.stack([[x.mean().item(), x.abs().mean().item()] for x in d["encoder"]])
    assert np.abs(es - g["enc_stats"]).max() < 2e-2 * np.abs(g["enc_stats"]).max()
    m.train()
    with torch.no_grad():
        tr = m(img)                     # train mode returns the (cls, dist) tuple, de_vit.py:324-325
        assert isinstance(tr, tuple) and len(tr) == 2
        from devit_amd import engine    # golden train outputs were captured with drop_path = 0
        tr = engine._forward_with_dp(m, img, None)["output"]
    assert chk(rel(tr[0], g["train_cls"]), 2e-2) and chk(rel(tr[1], g["train_dist"]), 2e-2)
    m.eval() if which == "deitb" else None


def test_modules_vs_golden(golden, models, dev):
    s, t, _, _ = models
    for tag, m, D, H in (("S", s, 384, 6), ("T", t, 768, 12)):
        g = golden(f"module_{tag}")
        blk = m.blocks[5]
        blk.eval()
        x = torch.from_numpy(det_array(f"x/{tag}", (2, 198, D))).to(dev).requires_grad_(True)
        sub = lambda z: z[:, ::9]
        for p in blk.parameters():
            p.grad = None
        y = blk.mlp(x)
        assert chk(rel(sub(y), g["mlp_y"]), 1e-2)
        y.square().sum().backward()
        assert chk(rel(sub(x.grad), g["mlp_dx"]), 1.5e-2)
        assert chk(rel(blk.mlp.fc1.weight.grad[:8], g["mlp_dw1_rows"]), 1.5e-2)
        blk.mlp.gate = torch.from_numpy(g["neuron_gate"])
        with torch.no_grad():
            y2 = blk.mlp(x)
        assert chk(rel(sub(y2), g["mlp_y_gated"]), 1e-2)
        assert chk(rel(blk.mlp.neuron_output.float().sum(dim=(0, 1)), g["mlp_neuron_output_sum"]), 1e-2)   # post-mask (Q3)
        blk.mlp.gate = torch.ones(4 * D)
        x.grad = None
        a = blk.attn(x, True)
        assert chk(rel(sub(a["output"]), g["attn_y"]), 1e-2)
        assert chk(rel(a["qkv"][0][:, :, :16], g["attn_q"]), 1e-2) and chk(rel(a["qkv"][2][:, :, :16], g["attn_v"]), 1e-2)
        a["output"].square().sum().backward()
        assert chk(rel(sub(x.grad), g["attn_dx"]), 1.5e-2)
        blk.attn.gate = torch.from_numpy(g["head_gate"])
        with torch.no_grad():
            a2 = blk.attn(x, True)
        assert chk(rel(sub(a2["output"]), g["attn_y_gated"]), 1e-2)
        assert chk(rel(blk.attn.head_output.float().sum(dim=(0, 1)), g["attn_head_output_sum"]), 1e-2)
        blk.attn.gate = torch.ones(H)
        with torch.no_grad():
            b = blk(x, output_qkv=True, output_att=True)
        assert chk(rel(sub(b["output"]), g["block_y"]), 1e-2) and chk(rel(sub(b["attention"]), g["block_att"]), 1e-2)
        for p in blk.parameters():
            p.grad = None
    s.train()


def test_losses_vs_golden(golden, dev):
    import devit_amd
    g = golden("loss_cls")
    lo = torch.from_numpy(det_array("lo", (8, C), std=1.5)).to(dev).requires_grad_(True)
    lk = torch.from_numpy(det_array("lk", (8, C), std=1.5)).to(dev).requires_grad_(True)
    lt = torch.from_numpy(det_array("lt", (8, C), std=2.0)).to(dev)
    soft = torch.from_numpy(g["soft_targets"]).to(dev)
    for kind, tau in (("hard", 1.0), ("soft", 3.0)):
        crit = devit_amd.DistillLoss(devit_amd.SoftTargetCrossEntropy(), kind, 0.5, tau)
        l = crit((lo, lk), lt, soft)
        d = torch.autograd.grad(l, [lo, lk])
        assert abs(float(l) - float(g[f"{kind}_loss"])) < 1e-5 * abs(float(g[f"{kind}_loss"]))   # fp32 kernel
        assert chk(rel(d[0], g[f"{kind}_dlo"]), 1e-5) and chk(rel(d[1], g[f"{kind}_dlk"]), 1e-5)
    # hard labels with the two base criteria distill_sub.py:345-352 picks when mixup is off
    g = golden("loss_cls_hardlabels")
    y = torch.from_numpy(g["labels"]).to(dev)
    for bname, base in (("ls", devit_amd.LabelSmoothingCrossEntropy(smoothing=0.1)), ("ce", torch.nn.CrossEntropyLoss())):
        for kind, tau in (("none", 1.0), ("hard", 1.0), ("soft", 3.0)):
            l = devit_amd.DistillLoss(base, kind, 0.5, tau)((lo, lk), lt, y)
            d = torch.autograd.grad(l, [lo, lk], allow_unused=True)
            assert abs(float(l) - float(g[f"{bname}_{kind}_loss"])) < 1e-5 * abs(float(g[f"{bname}_{kind}_loss"])), (bname, kind)
            assert chk(rel(d[0], g[f"{bname}_{kind}_dlo"]), 1e-5)
            if kind != "none":
                assert chk(rel(d[1], g[f"{bname}_{kind}_dlk"]), 1e-5)
    with pytest.raises(ValueError):       # the reference's LabelSmoothingCrossEntropy gathers by int labels (utils/losses.py:23)
        devit_amd.DistillLoss(devit_amd.LabelSmoothingCrossEntropy(0.1), "hard", 0.5, 1.0)((lo, lk), lt, soft)
    g = golden("loss_relation")
    tf = torch.from_numpy(det_array("tf", (2, 198, 3, 12, 64), std=0.25)).to(dev).permute(2, 0, 3, 1, 4)[1]
    sf = torch.from_numpy(det_array("sf", (2, 198, 3, 6, 64), std=0.25)).to(dev).permute(2, 0, 3, 1, 4)[1]
    sf = sf.detach().requires_grad_(True)
    l = devit_amd.feature_relation_loss(tf, sf)
    d, = torch.autograd.grad(l, [sf])
    assert abs(float(l) - float(g["loss"])) < 2e-2 * abs(float(g["loss"]))        # bf16 features
    assert chk(rel(d[:, :, ::9], g["dstudent"]), 1.5e-2)


def test_distill_step_vs_golden(golden, models, dev):
    from devit_amd import engine
    s, t, _, _ = models
    g = golden("step_bs8")
    img = torch.from_numpy(det_array("img8", (8, 3, 224, 224))).to(dev)
    soft = torch.from_numpy(g["soft_targets"]).to(dev)
    dps = torch.from_numpy(g["dp_scales"]).to(dev)
    s.train()
    for p in s.parameters():
        p.grad = None
    out = engine.distill_forward(s, t, img, soft, gama=(0.2, 0.1, 0.3), kind="hard", alpha=0.5, tau=1.0,
                                 dp_scales=[(dps[i, 0].contiguous(), dps[i, 1].contiguous()) for i in range(12)])
    for k in ("loss", "cls_loss", "q_loss", "k_loss", "v_loss"):
        e = abs(float(out[k]) - float(g[k])) / abs(float(g[k]))
        assert chk(e, 5e-4), f"{k}: {float(out[k])} vs {float(g[k])}"           # measured <= 8e-5
    assert chk(rel(out["teacher_logits"], g["teacher_logits"]), 1.5e-2)
    out["loss"].backward()
    with open(os.path.join(os.path.dirname(__file__), "golden", "step_param_names.json")) as f:
        names = json.load(f)
    params = dict(s.named_parameters())
    gn = np.array([params[n].grad.norm().item() for n in names])
    chk(float((np.abs(gn - g["grad_norms"]) / (g["grad_norms"] + 1e-3 * g["grad_norms"].max())).max()), 3e-3)
    bad = np.abs(gn - g["grad_norms"]) > 3e-3 * g["grad_norms"] + 3e-6 * g["grad_norms"].max()      # measured 6e-4
    assert not bad.any(), [(names[i], gn[i], g["grad_norms"][i]) for i in np.nonzero(bad)[0][:8]]
    assert chk(rel(params["head.weight"].grad, g["g_head_w"]), 1.5e-2)
    assert chk(rel(params["blocks.5.attn.qkv.weight"].grad[::48], g["g_qkv5_w_rows"]), 1.5e-2)
    assert chk(rel(params["blocks.0.mlp.fc1.weight"].grad[::64], g["g_fc1_0_rows"]), 1.5e-2)
    assert chk(rel(params["blocks.11.mlp.fc2.weight"].grad[::16], g["g_fc2_11_rows"]), 1.5e-2)
    assert chk(rel(params["pos_embed"].grad[0, ::8], g["g_pos"]), 1.5e-2)
    assert chk(rel(params["cls_token"].grad, g["g_cls"]), 1.5e-2)
    assert chk(rel(params["patch_embed.proj.weight"].grad[::16].reshape(-1, 768), g["g_patch_w"]), 1.5e-2)
    assert chk(rel(params["norm.weight"].grad, g["g_norm_w"]), 1.5e-2)
    for p in s.parameters():
        p.grad = None


def test_fresh_inputs_vs_oracle(models, dev):
    """Seeded inputs that are NOT in the fixture set, bs 4, student eval forward vs the CPU oracle."""
    s, _, st_s, _ = models
    img = torch.from_numpy(det_array("fresh", (4, 3, 224, 224), std=0.7))
    with torch.no_grad():
        ref = O.forward(st_s, GS, img, training=False)["output"]
    s.eval()
    with torch.no_grad():
        out = s(img.to(dev))
    assert chk(rel(out, ref.numpy()), 2e-2)
    assert torch.equal(out.argmax(1).cpu(), ref.argmax(1))
    s.train()


# ------------------------------------------------------------------------------------------ exact-fp32 parity path
def test_f32_path_meets_1e3_bar(golden, models, dev):
    """BASELINE.json north_star: logits within 1e-3 rel of the reference, top-1 bit-exact.  precision="f32" runs the
    same graph on the fp32 entry points (csrc/sgemm.hip); measured deviation is ~1e-5."""
    from devit_amd import engine
    s, t, _, _ = models
    img = torch.from_numpy(det_array("img8", (8, 3, 224, 224))).to(dev)
    try:
        s.precision = t.precision = "f32"
        for which, m in (("dedeit", s), ("deitb", t)):
            g = golden(f"model_{which}")
            m.eval()
            with torch.no_grad():
                logits = m(img)
                d = m(img, output_qkv=True, output_att=True)
            assert chk(rel(logits, g["logits"]), 1e-3), rel(logits, g["logits"])
            assert np.array_equal(logits.argmax(1).cpu().numpy(), g["top1"])
            assert chk(rel(d["qkv"][5][0][:2, :, :24], g["q5"]), 1e-3) and chk(rel(d["attention"][5][:2, :24], g["att5"]), 1e-3)
        t.eval()
        s.train()
        g = golden("step_bs8")
        for p in s.parameters():
            p.grad = None
        dps = torch.from_numpy(g["dp_scales"]).to(dev)
        out = engine.distill_forward(s, t, img, torch.from_numpy(g["soft_targets"]).to(dev),
                                     dp_scales=[(dps[i, 0].contiguous(), dps[i, 1].contiguous()) for i in range(12)])
        for k in ("loss", "cls_loss", "q_loss", "k_loss", "v_loss"):
            assert abs(float(out[k]) - float(g[k])) < 1e-4 * abs(float(g[k])), (k, float(out[k]), float(g[k]))
        out["loss"].backward()
        with open(os.path.join(os.path.dirname(__file__), "golden", "step_param_names.json")) as f:
            names = json.load(f)
        params = dict(s.named_parameters())
        gn = np.array([params[n].grad.norm().item() for n in names])
        assert np.abs(gn - g["grad_norms"]).max() < 1e-3 * g["grad_norms"].max()
        assert chk(rel(params["blocks.5.attn.qkv.weight"].grad[::48], g["g_qkv5_w_rows"]), 1e-3)
        assert chk(rel(params["patch_embed.proj.weight"].grad[::16].reshape(-1, 768), g["g_patch_w"]), 1e-3)
        assert chk(rel(params["pos_embed"].grad[0, ::8], g["g_pos"]), 1e-3)
    finally:
        s.precision = t.precision = "bf16"
        for p in s.parameters():
            p.grad = None


# ------------------------------------------------------------------------------------------ ensemble stage (config 5)
def test_ensemble_vs_golden(golden, dev):
    """MultiViT(4 x dedeit) + EnsMLP + EnsLoss against the reference's own modules (models/ensemble_models.py,
    utils/losses.py:180-244), including the positional checkpoint copy of ensemble.py:192-200."""
    import devit_amd
    from devit_amd import engine, losses
    from devit_amd.ensemble_models import EnsMLP, MultiViT, load_sub_checkpoints
    g = golden("ensemble")
    multi = MultiViT("dedeit", drop=0, drop_path=0.0, num_classes_list=[25] * 4, num_div=4)
    assert len(multi.state_dict()) == int(g["n_multi_keys"])
    load_sub_checkpoints(multi, [O.make_state(GS, 25, f"E{i}") for i in range(4)])
    ens = EnsMLP("dedeit", 100, 384, [25] * 4, 768)
    ens.load_state_dict(O.make_ens_state())
    teacher = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=100)
    teacher.load_state_dict(O.make_state(GT, 100, "T100"))
    multi.to(dev); ens.to(dev); teacher.to(dev).eval()
    img = torch.from_numpy(det_array("img4", (4, 3, 224, 224))).to(dev)
    multi.eval(); ens.eval()
    with torch.no_grad():
        le = ens(multi(img))
    assert chk(rel(le, g["logits_eval"]), 2e-2) and np.array_equal(le.argmax(1).cpu().numpy(), g["logits_eval"].argmax(1))
    multi.train(); ens.train()
    crit = losses.EnsLoss(losses.SoftTargetCrossEntropy(), teacher, "dedeit", "hard", 0.5, 1.0)
    out = engine.ens_forward(multi, ens, crit, img, torch.from_numpy(g["soft_targets"]).to(dev))
    assert abs(float(out["token_loss"]) - float(g["token_loss"])) < 2e-2 * float(g["token_loss"])
    assert abs(float(out["cls_loss"]) - float(g["cls_loss"])) < 2e-2 * float(g["cls_loss"])
    out["loss"].backward()
    assert chk(rel(ens.cls_mlp.weight.grad[::48], g["g_cls_mlp_w_rows"]), 2.5e-2)
    assert chk(rel(ens.dist_classifier.weight.grad[::10], g["g_dist_cls_w"]), 2.5e-2)
    assert chk(rel(multi.backbones[2].blocks[3].mlp.fc1.weight.grad[::96], g["g_b2_fc1_rows"]), 2.5e-2)
    assert chk(rel(multi.backbones[0].pos_embed.grad[0, ::16], g["g_b0_pos"]), 2.5e-2)
    # exact-fp32 path on the same graph: the 1e-3 bar
    for m in list(multi.backbones) + [teacher]:
        m.precision = "f32"
    multi.eval(); ens.eval()
    with torch.no_grad():
        le = ens(multi(img))
    assert chk(rel(le, g["logits_eval"]), 1e-3)


# ------------------------------------------------------------------------------------------ physical shrinking (§8f-2)
def test_compacted_model_equals_masked_model(models, dev):
    """shrink.compact(): the physically shrunk student (masked heads / neurons removed from the GEMMs and the attention
    grid) computes the masked model's function -- against the masked HIP model and against the CPU oracle with gates."""
    from devit_amd import _lib, shrink
    s, _, st_s, _ = models
    g = torch.Generator().manual_seed(7)
    head_gates, neuron_gates = [], []
    for i, blk in enumerate(s.blocks):
        hm = torch.ones(6)
        hm[torch.randperm(6, generator=g)[: (1 if i % 3 == 0 else 2)]] = 0          # 5 kept (-> 6 run) or 4 kept
        nm = torch.ones(1536)
        nm[torch.randperm(1536, generator=g)[:461]] = 0                              # shrink_ratio 0.3
        blk.attn.gate, blk.mlp.gate = hm, nm
        head_gates.append(hm)
        neuron_gates.append(nm)
    img = torch.from_numpy(det_array("shrink", (4, 3, 224, 224), std=0.7))
    try:
        s.eval()
        with torch.no_grad():
            ref = O.forward(st_s, GS, img, training=False, head_gates=head_gates, neuron_gates=neuron_gates)["output"]
            masked = s(img.to(dev), output_qkv=True)
            rep = shrink.compact(s)
            assert all(hr in (4, 6) and nr == 1152 for _, hr, _, nr in rep)
            dense_gf, compact_gf = 9.247, shrink.compacted_gflops(s, num_classes=C)
            assert 0.70 * dense_gf < compact_gf < 0.80 * dense_gf
            comp = s(img.to(dev), output_qkv=True)
        assert chk(rel(comp["output"], masked["output"].float().cpu().numpy()), 1e-2)      # same bf16 path, fewer zero terms
        assert chk(rel(comp["output"], ref.numpy()), 2e-2) and torch.equal(comp["output"].argmax(1).cpu(), ref.argmax(1))
        assert comp["qkv"][0][0].shape[1] == rep[0][1]                                 # q of block 0: [B, heads run, N, 64]
        assert s.blocks[0].mlp.neuron_output.shape[-1] == 1152
        s.train()
        with pytest.raises(_lib.DevitError):                                           # inference-only
            s(img.to(dev))
        shrink.uncompact(s)
        s.eval()
        with torch.no_grad():
            again = s(img.to(dev))
        assert torch.equal(again, masked["output"])
    finally:
        shrink.uncompact(s)
        for blk in s.blocks:
            blk.attn.gate, blk.mlp.gate = torch.ones(6), torch.ones(1536)
        s.train()


def test_training_through_compacted_blocks(golden, models, dev):
    """distill_sub.py:384-401 ranks one batch, sets the gates and then TRAINS the gated student; core/imp_rank.py:65-71,
    147-153 only mask, so the reference pays the dense model's FLOPs there.  shrink.compact(model, trainable=True) trains
    through the compacted blocks: one DEKD step (bs 8, recorded DropPath masks, 0.3 / 0.3 gates) against the masked HIP
    path and against the oracle with the same gates -- losses, every parameter's gradient norm, exact zeros on the masked
    units, and an optimizer step that moves the compact weights."""
    from devit_amd import engine, shrink
    s, t, st_s, st_t = models
    g = golden("step_bs8")
    img_c = torch.from_numpy(det_array("img8", (8, 3, 224, 224)))
    img = img_c.to(dev)
    soft_c = torch.from_numpy(g["soft_targets"])
    soft = soft_c.to(dev)
    dps_c = torch.from_numpy(g["dp_scales"])
    dps = [(dps_c[i, 0].contiguous().to(dev), dps_c[i, 1].contiguous().to(dev)) for i in range(12)]
    gen = torch.Generator().manual_seed(17)
    head_gates, neuron_gates = [], []
    for i in range(12):
        hm, nm = torch.ones(6), torch.ones(1536)
        hm[torch.randperm(6, generator=gen)[:2]] = 0              # int(6 * 0.7) = 4 heads kept
        nm[torch.randperm(1536, generator=gen)[:461]] = 0         # int(1536 * 0.7) = 1075 neurons kept
        head_gates.append(hm)
        neuron_gates.append(nm)

    def step():
        for p in s.parameters():
            p.grad = None
        out = engine.distill_forward(s, t, img, soft, gama=(0.2, 0.1, 0.3), kind="hard", alpha=0.5, tau=1.0, dp_scales=dps)
        out["loss"].backward()
        torch.cuda.synchronize()
        return ({k: float(out[k]) for k in ("loss", "cls_loss", "q_loss", "k_loss", "v_loss")},
                {n: p.grad.detach().clone() for n, p in s.named_parameters()})
    s.train()
    try:
        for blk, hm, nm in zip(s.blocks, head_gates, neuron_gates):
            blk.attn.gate, blk.mlp.gate = hm.clone(), nm.clone()
        lm, gm = step()                                            # masked: dense FLOPs
        rep = shrink.compact(s, trainable=True)
        assert [r[1] for r in rep] == [4] * 5 + [6] + [4] * 6      # block 5 feeds the relation loss: all heads stay
        assert all(r[3] == 1152 for r in rep)
        lc, gc = step()                                            # compacted
        ref_state = {k: v.clone().requires_grad_(True) for k, v in st_s.items()}
        ref = O.distill_step(ref_state, GS, st_t, GT, img_c, soft_c, dp_scales=dps_c, head_gates=head_gates,
                             neuron_gates=neuron_gates)
        for k in lm:
            assert chk(abs(lc[k] - lm[k]) / abs(lm[k]), 2e-3), (k, lc[k], lm[k])            # same bf16 kernels, fewer zero terms
            assert chk(abs(lc[k] - float(ref[k])) / abs(float(ref[k])), 2e-3), (k, lc[k], float(ref[k]))
        ref["loss"].backward()
        names = list(gm)
        nm_ = torch.stack([gm[n].norm() for n in names]).cpu()
        nc_ = torch.stack([gc[n].norm() for n in names]).cpu()
        bad = (nc_ - nm_).abs() > 1e-2 * nm_ + 1e-3 * nm_.max()
        chk(float(((nc_ - nm_).abs() / (nm_ + 1e-3 * nm_.max())).max()), 1e-2)
        assert not bool(bad.any()), [(n, float(a), float(b)) for n, a, b, f in zip(names, nc_, nm_, bad) if f][:8]
        for i, (hm, nmk) in enumerate(zip(head_gates, neuron_gates)):
            dead_n = (nmk == 0).to(dev)
            for gsel in (gm, gc):
                assert float(gsel[f"blocks.{i}.mlp.fc1.weight"][dead_n].abs().max()) == 0.0       # masked neurons: exact zeros
                assert float(gsel[f"blocks.{i}.mlp.fc1.bias"][dead_n].abs().max()) == 0.0
                assert float(gsel[f"blocks.{i}.mlp.fc2.weight"][:, dead_n].abs().max()) == 0.0
                dead_h = (hm == 0).to(dev)
                pw = gsel[f"blocks.{i}.attn.proj.weight"].view(384, 6, 64)[:, dead_h]
                assert float(pw.abs().max()) == 0.0
                qw = gsel[f"blocks.{i}.attn.qkv.weight"].view(3, 6, 64, 384)[:, dead_h]
                assert (float(qw.abs().max()) == 0.0) == (i != 5)       # the relation loss reaches the masked heads of block 5
        # every gradient norm against the oracle's (fp32 CPU, same gates)
        nr_ = torch.stack([ref_state[n].grad.norm() for n in names])
        chk(float(((nc_ - nr_).abs() / (nr_ + 1e-3 * nr_.max())).max()), 1e-2)
        bad = (nc_ - nr_).abs() > 1e-2 * nr_ + 1e-3 * nr_.max()
        assert not bool(bad.any()), [(n, float(a), float(b)) for n, a, b, f in zip(names, nc_, nr_, bad) if f][:8]
        # one optimizer step moves the masters; the next forward re-gathers the compact weights from them
        before = s.blocks[2]._compact["fc1_w16"].clone()
        with torch.no_grad():
            for n, p in s.named_parameters():
                p.add_(gc[n], alpha=-1.0)
        for p in s.parameters():
            p.grad = None
        out = engine.distill_forward(s, t, img, soft, dp_scales=dps)
        assert not torch.equal(s.blocks[2]._compact["fc1_w16"], before)
        assert float(out["loss"]) == float(out["loss"])
        with torch.no_grad():
            for n, p in s.named_parameters():
                p.add_(gc[n], alpha=1.0)
        # eval mode, weights swapped in by `p.data = other` (how EMA weights are usually evaluated; it need not bump the version counter):
        # the compact copies must follow the storage, not only the counter (advisor r04)
        s.eval()
        with torch.no_grad():
            s(img)                                                   # settles the eval-mode gather
            w = s.blocks[2].mlp.fc1.weight
            old_data, seen = w.data, s.blocks[2]._compact["fc1_w16"].clone()
            w.data = old_data * 1.5
            s(img)
            assert not torch.equal(s.blocks[2]._compact["fc1_w16"], seen), "compact fc1 weights did not follow a `.data =` swap"
            w.data = old_data
            s(img)
            assert torch.equal(s.blocks[2]._compact["fc1_w16"], seen)
    finally:
        shrink.uncompact(s)
        for blk in s.blocks:
            blk.attn.gate, blk.mlp.gate = torch.ones(6), torch.ones(1536)
        for p in s.parameters():
            p.grad = None
        s.train()


# ------------------------------------------------------------------------------------------ non-distilled geometry
def test_gate_reassignment_invalidates_block_cache(models, dev):
    """The reference's shrink pattern (core/imp_rank.py:65-79): forward with mask A, `m.gate = ones`, `m.gate = mask B`,
    forward.  The cached BlockParams must follow every assignment (CPython hands a freed tensor's id to the next one, so
    the cache key is a version counter, never id()), and an in-place edit of the gate tensor too."""
    s, _, _, _ = models
    s.eval()
    img = torch.from_numpy(det_array("img8", (8, 3, 224, 224)))[:2].to(dev)
    blk = s.blocks[3]
    H, Hd = blk.attn.num_heads, blk.mlp.hidden_features

    def run():
        with torch.no_grad():
            return s(img).clone()

    def fresh():
        for b in s.blocks:
            b.__dict__.pop("_bp_cache", None)
        return run()
    dense = run()
    mask_a, mask_b = torch.ones(Hd), torch.ones(Hd)
    mask_a[: Hd // 2] = 0
    mask_b[Hd // 3:] = 0
    try:
        blk.mlp.gate = mask_a
        out_a = run()
        blk.mlp.gate = torch.ones(Hd)
        blk.mlp.gate = mask_b.clone()
        out_b = run()
        assert not torch.equal(out_a, out_b) and not torch.equal(out_b, dense)
        assert torch.equal(out_b, fresh())
        blk.mlp.gate[: Hd // 3] = 0            # in place, no assignment: everything masked now
        out_c = run()
        assert not torch.equal(out_c, out_b) and torch.equal(out_c, fresh())
        hm = torch.ones(H)
        hm[0] = 0
        blk.attn.gate = hm
        out_d = run()
        blk.attn.gate = torch.ones(H)
        blk.attn.gate = torch.ones(H)
        assert not torch.equal(out_d, out_c)
    finally:
        blk.mlp.gate = torch.ones(Hd)
        blk.attn.gate = torch.ones(H)
    assert torch.equal(run(), dense)
    s.train()


def test_nondistilled_devit_vs_oracle(dev):
    """`devit` (one class token, 197 token rows per image, tensor output in both modes; models/de_vit.py:495-513) at a
    ragged batch (3 images -> 591 rows, padded to 768): eval logits and a full backward against the CPU oracle's
    autograd."""
    import devit_amd
    geom = O.GEOMETRY["devit"]
    st = O.make_state(geom, C, "V")
    m = devit_amd.create_model("devit", num_classes=C, drop_path_rate=0.0).to(dev)
    m.load_state_dict(st)
    img = torch.from_numpy(det_array("devit", (3, 3, 224, 224), std=0.7))
    y = torch.tensor([3, 17, 8])
    st_g = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    ref = O.forward(st_g, geom, img, training=True)["output"]
    assert isinstance(ref, torch.Tensor) and ref.shape == (3, C)
    torch.nn.functional.cross_entropy(ref, y).backward()
    m.eval()
    with torch.no_grad():
        out_eval = m(img.to(dev))
    assert chk(rel(out_eval, ref.detach().numpy()), 2e-2) and torch.equal(out_eval.argmax(1).cpu(), ref.argmax(1))
    m.train()
    out = m(img.to(dev))
    assert isinstance(out, torch.Tensor)
    torch.nn.functional.cross_entropy(out.float(), y.to(dev)).backward()
    for name, p in m.named_parameters():
        g_ref = st_g[name].grad
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
        assert chk(rel(p.grad, g_ref.numpy()), 3e-2), (name, rel(p.grad, g_ref.numpy()))


def test_vit_large_teacher_forward_vs_oracle(dev):
    """`vit_large_patch16_224` -- the DEFAULT --teacher-model of distill_sub.py:141 (models/deit_vit.py:517-525: 1024 wide, 24
    blocks, 16 heads, one class token) -- through the HIP path at bs 2 against the CPU oracle: eval logits, top-1, and the
    q / k / v views of the block DEKD would read (teacher_layer_num // 2 - 1 = 11, engine.py:91-92)."""
    import devit_amd
    geom = O.GEOMETRY["vit_large_patch16_224"]
    st = O.make_state(geom, C, "L")
    m = devit_amd.create_model("vit_large_patch16_224", num_classes=C).to(dev).eval()
    m.load_state_dict(st)
    assert len(m.blocks) == 24 and m.embed_dim == 1024 and m.blocks[0].attn.num_heads == 16
    img = torch.from_numpy(det_array("vit_large", (2, 3, 224, 224), std=0.7))
    with torch.no_grad():
        ref = O.forward(st, geom, img, training=False)
        out = m(img.to(dev), output_qkv=True)
    assert isinstance(out["output"], torch.Tensor) and out["output"].shape == (2, C)
    assert chk(rel(out["output"], ref["output"].numpy()), 2e-2)
    assert torch.equal(out["output"].argmax(1).cpu(), ref["output"].argmax(1))
    assert len(out["qkv"]) == 24
    for got, want in zip(out["qkv"][11], ref["qkv"][11]):
        assert got.shape == (2, 16, 197, 64) and chk(rel(got, want.numpy()), 2e-2)
    with torch.no_grad():                       # the plain call: tensor, not dict (models/de_vit.py:316-325)
        plain = m(img.to(dev))
    assert torch.equal(plain, out["output"])


@pytest.mark.parametrize("name", ["deit_tiny_patch16_224", "deit_tiny_distilled_patch16_224", "vit_tiny_patch16_224"])
def test_narrow_names_forward_vs_oracle(dev, name):
    """The three D = 192 names of models/deit_vit.py:457-525 (3 heads, hidden 768): not a geometry of the MFMA tiles, so they are pinned to the
    exact-fp32 kernels (de_vit.check_geometry) -- and therefore held to the fp32 bar: eval logits and q / k / v of block 5 within 1e-4 of the
    CPU oracle, top-1 exact, the return-type matrix of models/de_vit.py:316-334."""
    import devit_amd
    geom = O.GEOMETRY[name]
    st = O.make_state(geom, C, "N")
    m = devit_amd.create_model(name, num_classes=C).to(dev).eval()
    m.load_state_dict(st)
    assert m.precision == "f32" and m.embed_dim == 192 and [k for k in m.state_dict()] == list(st)
    img = torch.from_numpy(det_array("narrow", (4, 3, 224, 224), std=0.7))
    ntok = 2 if geom["distilled"] else 1
    with torch.no_grad():
        ref = O.forward(st, geom, img, training=False)
        out = m(img.to(dev), output_qkv=True)
        plain = m(img.to(dev))
    assert out["output"].shape == (4, C) and torch.equal(plain, out["output"])
    assert chk(rel(out["output"], ref["output"].numpy()), 1e-4)
    assert torch.equal(out["output"].argmax(1).cpu(), ref["output"].argmax(1))
    for got, want in zip(out["qkv"][5], ref["qkv"][5]):
        assert got.shape == (4, 3, 196 + ntok, 64) and chk(rel(got, want.numpy()), 1e-4)


def test_narrow_student_distill_step_vs_oracle(dev):
    """One DEKD step (engine.py:68-127) with `deit_tiny_distilled_patch16_224` as the STUDENT under the DeiT-B teacher's bf16 kernels, bs 8,
    recorded DropPath masks: five losses within 2e-2 of the oracle (the teacher side carries bf16 rounding: test_distill_step_vs_golden's bar),
    every gradient norm within 2e-2.  With the teacher on the fp32 path too (precision="f32"): losses 1e-4, gradient norms 1e-3."""
    import devit_amd
    from devit_amd import engine
    gs, gt = O.GEOMETRY["deit_tiny_distilled_patch16_224"], O.GEOMETRY["deit_base_distilled_patch16_224"]
    st_s, st_t = O.make_state(gs, C, "NS"), O.make_state(gt, C, "T")
    s = devit_amd.create_model("deit_tiny_distilled_patch16_224", num_classes=C, drop_path_rate=0.1, drop_block_rate=None)
    t = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=C)
    s.load_state_dict(st_s)
    t.load_state_dict(st_t)
    s.to(dev).train()
    t.to(dev).eval()
    for p in t.parameters():
        p.requires_grad_(False)
    img = torch.from_numpy(det_array("narrow_step", (8, 3, 224, 224), std=0.8))
    g = torch.Generator().manual_seed(11)
    y = torch.randint(0, C, (8,), generator=g)
    soft = torch.full((8, C), 0.1 / C).scatter_(1, y[:, None], 0.9 + 0.1 / C)
    keep = 1.0 - torch.linspace(0, 0.1, 12)
    sc = torch.floor(keep.view(12, 1, 1) + torch.rand((12, 2, 8), generator=g)) / keep.view(12, 1, 1)
    masks = [(sc[i, 0].contiguous(), sc[i, 1].contiguous()) for i in range(12)]
    params = {k: v.clone().requires_grad_(True) for k, v in st_s.items()}
    ref = O.distill_step(params, gs, st_t, gt, img, soft, dp_scales=masks)
    ref["loss"].backward()
    names = list(params)
    nref = torch.stack([params[n].grad.norm() for n in names])
    for tprec, lbar, gbar in (("bf16", 2e-2, 2e-2), ("f32", 1e-4, 1e-3)):
        t.precision = tprec
        for p in s.parameters():
            p.grad = None
        out = engine.distill_forward(s, t, img.to(dev), soft.to(dev), gama=(0.2, 0.1, 0.3), kind="hard", alpha=0.5, tau=1.0,
                                     dp_scales=[(a.to(dev), b.to(dev)) for a, b in masks])
        out["loss"].backward()
        torch.cuda.synchronize()
        for k in ("loss", "cls_loss", "q_loss", "k_loss", "v_loss"):
            e = abs(float(out[k].detach()) - float(ref[k].detach())) / abs(float(ref[k].detach()))
            assert chk(e, lbar), (tprec, k, float(out[k].detach()), float(ref[k].detach()))
        got = dict(s.named_parameters())
        nhip = torch.stack([got[n].grad.float().norm().cpu() for n in names])
        assert chk(float(((nhip - nref).abs() / (nref + 1e-3 * nref.max())).max()), gbar), tprec
    t.precision = "bf16"


# ------------------------------------------------------------------------------------------ batch-size edges
def test_batch_invariance_and_ragged_batches(models, dev):
    """Ragged inputs: batch sizes that fill no tile (1, 3, 19 images = 198 / 594 / 3762 token rows), changing from call
    to call (workspaces re-sized), train-mode steps in between.  No forward kernel mixes rows of different images, so
    image i of a batch must give the bits of the same image run alone; bs-1 also goes against the CPU oracle."""
    s, t, st_s, _ = models
    img = torch.from_numpy(det_array("ragged", (19, 3, 224, 224), std=0.7)).to(dev)
    s.eval()
    with torch.no_grad():
        full_s, full_t = s(img), t(img)
        for lo, hi in ((0, 1), (7, 10), (18, 19), (0, 19)):
            assert torch.equal(s(img[lo:hi]), full_s[lo:hi]), (lo, hi)
            assert torch.equal(t(img[lo:hi]), full_t[lo:hi]), (lo, hi)
        ref = O.forward(st_s, GS, img[:1].cpu(), training=False)["output"]
    assert chk(rel(full_s[:1], ref.numpy()), 2e-2)
    # a bs-3 training step (forward + backward) between two eval calls leaves the eval result untouched
    s.train()
    out = s(img[:3], output_qkv=True)
    logits = out["output"]
    loss = sum(o.float().square().mean() for o in (logits if isinstance(logits, tuple) else (logits,)))
    loss.backward()
    grads = [p.grad for p in s.parameters() if p.grad is not None]
    assert grads and all(bool(torch.isfinite(g).all()) for g in grads)
    s.zero_grad(set_to_none=True)
    s.eval()
    with torch.no_grad():
        assert torch.equal(s(img), full_s)
    s.train()


# ------------------------------------------------------------------------------------------ class counts
@pytest.mark.parametrize("classes", [10, 250, 1000])
def test_step_other_class_counts(dev, classes):
    """BASELINE configs[3] (ImageNet/4: 250 classes, what bench.py runs at N > 1) and the full 1000-class heads: the
    head GEMMs and the classification-loss kernel at widths other than the fixtures' 25.  bs-3 DEKD step vs the oracle."""
    import devit_amd
    from devit_amd import engine
    st_s, st_t = O.make_state(GS, classes, "S"), O.make_state(GT, classes, "T")
    s = devit_amd.create_model("dedeit", num_classes=classes, drop_path_rate=0.0, drop_block_rate=None)
    t = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=classes)
    s.load_state_dict(st_s)
    t.load_state_dict(st_t)
    s, t = s.to(dev).train(), t.to(dev).eval()
    img = torch.from_numpy(det_array("classes", (3, 3, 224, 224), std=0.7))
    soft = torch.softmax(torch.from_numpy(det_array("classes_soft", (3, classes), std=2.0)), 1)
    leaf = {k: v.clone().requires_grad_(True) for k, v in st_s.items()}
    ref = O.distill_step(leaf, GS, st_t, GT, img, soft)
    ref["loss"].backward()
    out = engine.distill_forward(s, t, img.to(dev), soft.to(dev), gama=(0.2, 0.1, 0.3), kind="hard", alpha=0.5, tau=1.0)
    out["loss"].backward()
    for k in ("loss", "cls_loss", "q_loss", "k_loss", "v_loss"):
        assert abs(float(out[k].detach()) - float(ref[k].detach())) < 2e-2 * abs(float(ref[k].detach())) + 1e-6, k
    assert chk(rel(out["logits"][0], ref["student"]["output"][0].detach().numpy()), 2e-2)
    assert chk(rel(out["teacher_logits"], ref["teacher"]["output"].numpy()), 2e-2)
    got = dict(s.named_parameters())
    for k in ("head.weight", "head_dist.weight", "head.bias", "blocks.11.mlp.fc2.weight", "blocks.0.attn.qkv.weight",
              "blocks.3.attn.qkv.bias", "blocks.7.mlp.fc1.bias",       # row sums fused into the weight-gradient GEMMs
              "blocks.7.mlp.fc2.bias", "blocks.2.attn.proj.bias",      # column sums fused into the LayerNorm backward
              "blocks.4.norm1.weight", "blocks.4.norm2.bias"):
        assert chk(rel(got[k].grad, leaf[k].grad.numpy()), 2.5e-2), k


# ------------------------------------------------------------------------------------------ composite block calls
@pytest.mark.parametrize("bs", [3, 8])
def test_block_calls_match_granular_path(models, dev, bs):
    """devit_encoder_fwd / devit_block_bwd (one host call per encoder / per block, csrc/encoder.hip) enqueue the same
    kernels with the same arguments in the same order as the one-call-per-kernel path: forward results and the
    input gradient are bit-identical, weight gradients agree to the order of their fp32 atomics.  bs 3 = 594 token rows:
    the zeroed pad rows (594 -> 768) of every buffer are the composite call's own job."""
    from devit_amd import engine, ops
    s, t, _, _ = models
    img = torch.from_numpy(det_array("img8", (8, 3, 224, 224)))[:bs].to(dev)
    soft = torch.softmax(torch.from_numpy(det_array("comp_soft", (bs, C), std=2.0)), 1).to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    dps = []
    for i in range(12):
        keep = 1.0 - 0.1 * i / 11
        sc = torch.floor(keep + torch.rand((2, bs), generator=g, device=dev)) / keep
        dps.append((sc[0].contiguous(), sc[1].contiguous()))
    res = {}
    s.train()
    try:
        for mode in (False, True):
            ops.COMPOSITE = mode
            for p in s.parameters():
                p.grad = None
            x = img.clone().requires_grad_(True)
            out = engine.distill_forward(s, t, x, soft, dp_scales=dps)
            out["loss"].backward()
            torch.cuda.synchronize()
            res[mode] = (out["loss"].detach().clone(), out["logits"][0].detach().clone(), out["teacher_logits"].detach().clone(),
                         {n: p.grad.detach().clone() for n, p in s.named_parameters()})
    finally:
        ops.COMPOSITE = True
        for p in s.parameters():
            p.grad = None
    (l0, lo0, tl0, g0), (l1, lo1, tl1, g1) = res[False], res[True]
    assert torch.equal(lo0, lo1) and torch.equal(tl0, tl1) and torch.equal(l0, l1)
    for n in g0:
        assert chk(rel(g1[n], g0[n].cpu().numpy()), 2e-5), n          # split-K atomics: summation order only


def test_weight_gradients_on_the_side_stream(models, dev, monkeypatch):
    """devit_block_bwd runs its four weight-gradient launches on a side stream of its own behind events of their producers and joins
    it before returning (csrc/encoder.hip).  Against DEVIT_WGRAD_STREAM=0 (everything on the caller's stream): loss and logits
    bit-identical, every parameter's gradient to the order of their fp32 atomics -- with every transient buffer of the caching
    allocator poisoned between the two runs, so that a weight-gradient launch that ran after its operands were recycled (a missing
    join) or before they were written (a missing event) would read NaNs."""
    from devit_amd import engine
    s, t, _, _ = models
    bs = 8
    img = torch.from_numpy(det_array("img8", (8, 3, 224, 224)))[:bs].to(dev)
    soft = torch.softmax(torch.from_numpy(det_array("comp_soft", (bs, C), std=2.0)), 1).to(dev)
    g = torch.Generator(device=dev).manual_seed(11)
    dps = []
    for i in range(12):
        keep = 1.0 - 0.1 * i / 11
        sc = torch.floor(keep + torch.rand((2, bs), generator=g, device=dev)) / keep
        dps.append((sc[0].contiguous(), sc[1].contiguous()))
    res = {}
    s.train()
    try:
        for flag in ("0", "1", "1"):
            monkeypatch.setenv("DEVIT_WGRAD_STREAM", flag)
            junk = [torch.full((n,), float("nan"), device=dev) for n in (1 << 24, 1 << 22, 1 << 20, 1 << 20)]
            del junk                                   # cached blocks the next run's transient buffers are carved from
            for p in s.parameters():
                p.grad = None
            x = img.clone().requires_grad_(True)
            out = engine.distill_forward(s, t, x, soft, dp_scales=dps)
            out["loss"].backward()
            res.setdefault(flag, []).append((out["loss"].detach().clone(), out["logits"][0].detach().clone(),
                                             {n: p.grad.detach().clone() for n, p in s.named_parameters()}))
        torch.cuda.synchronize()
    finally:
        for p in s.parameters():
            p.grad = None
    l0, lo0, g0 = res["0"][0]
    for l1, lo1, g1 in res["1"]:
        assert torch.equal(l0, l1) and torch.equal(lo0, lo1)
        for n in g0:
            assert bool(torch.isfinite(g1[n]).all()), n
            assert chk(rel(g1[n], g0[n].cpu().numpy()), 2e-5), n


# ------------------------------------------------------------------------------------------ DeiT criterion of train_subdata.py
def test_distillation_loss_vs_golden(golden, dev):
    """losses.DistillationLoss against the reference's own class (utils/losses.py:44-119, teacher inside the criterion)
    for every base criterion train_subdata.py:409-416 can select -- CrossEntropyLoss, LabelSmoothingCrossEntropy(0.1),
    SoftTargetCrossEntropy -- times none / hard / soft(tau = 3): loss and both logit gradients, fp32 kernel."""
    from devit_amd import losses
    g = golden("distillation_loss")
    lt = torch.from_numpy(g["lt"]).to(dev)

    class Teacher(torch.nn.Module):
        def forward(self, x, *a):
            return lt
    y, soft = torch.from_numpy(g["y"]).to(dev), torch.from_numpy(g["soft"]).to(dev)
    bases = {"ce": (torch.nn.CrossEntropyLoss(), y), "ls": (losses.LabelSmoothingCrossEntropy(0.1), y),
             "soft": (losses.SoftTargetCrossEntropy(), soft)}
    for bname, (base, labels) in bases.items():
        for kind, tau in (("none", 1.0), ("hard", 1.0), ("soft", 3.0)):
            lo = torch.from_numpy(g["lo"]).to(dev).requires_grad_(True)
            lk = torch.from_numpy(g["lk"]).to(dev).requires_grad_(True)
            crit = losses.DistillationLoss(base, Teacher(), kind, 0.5, tau, False)
            loss = crit(inputs=torch.zeros(8, 3, 8, 8, device=dev), outputs=(lo, lk), labels=labels)
            dlo, dlk = torch.autograd.grad(loss, [lo, lk], allow_unused=True)
            ref = float(g[f"{bname}_{kind}_loss"])
            assert chk(abs(float(loss) - ref) / abs(ref), 1e-5), (bname, kind, float(loss), ref)
            assert chk(rel(dlo, g[f"{bname}_{kind}_dlo"]), 1e-5), (bname, kind)
            if kind != "none":
                assert chk(rel(dlk, g[f"{bname}_{kind}_dlk"]), 1e-5), (bname, kind)
            else:
                assert dlk is None or float(dlk.abs().max()) == 0.0


# ------------------------------------------------------------------------------------------ f16 frozen-teacher forward
# Measured on MI355X (round 2, first GPU run of this path): f16 teacher logits deviate 1.126e-3 of max|logit| from the
# reference on the 8-image golden -- 6x closer than bf16 (6.8e-3) but NOT inside BASELINE.json's 1e-3.  This is not a kernel
# defect: per block, the GPU residual stream tracks the CPU emulation of f16 storage (profiles/r02_b_f16_localise.txt: after
# block 11, 5.6e-4 from the fp32 reference on the GPU vs 6.4e-4 emulated), and the logit statistic (a max over rounding noise)
# moves by +-40 % between samples: emulation 7.0e-4 on 8 images but 1.02e-3 on 2, GPU 1.13e-3 on 8 but 8.2e-4 on 2.  f16
# storage lands AROUND 1e-3, straddling the bar; the earlier "7.0e-4 predicted" was one draw read as margin.  The 1e-3 claim
# therefore lives in a strict xfail below: it shows in every run, and turns red if the path ever gets under the bar so that
# the marker has to go.  Only precision="f32" (test_f32_path_meets_1e3_bar) meets 1e-3.
F16_LOGITS_MEASURED = 1.126e-3
F16_LOGITS_REGRESSION_BAR = 2.3e-3        # ~2x measured, as for the bf16 bars above


def _f16_teacher_logits(models, dev):
    from devit_amd import ops
    _, t, _, _ = models
    img = torch.from_numpy(det_array("img8", (8, 3, 224, 224))).to(dev)
    try:
        t.precision = "f16"
        t.eval()
        with torch.no_grad():
            return t(img)
    finally:
        ops.COMPOSITE = True
        t.precision = "bf16"


@pytest.mark.xfail(strict=True, reason="f16 teacher forward measured 1.126e-3 on MI355X: misses BASELINE.json's 1e-3; "
                                       "f16 storage lands around 1e-3 (kernels track the emulation per block, "
                                       "profiles/r02_b_f16_localise.txt)")
def test_f16_teacher_forward_meets_1e3_bar(golden, models, dev):
    """BASELINE.json's bar, unchanged, on the benchmarked kernels with f16 operands.  Expected to fail until the path is fixed."""
    e = rel(_f16_teacher_logits(models, dev), golden("model_deitb")["logits"])
    assert e <= 1e-3, f"f16 teacher logits rel-to-max err {e:.3e}"


def test_f16_teacher_forward(golden, models, dev):
    """precision="f16" (IEEE f16 MFMA operands and stored activations, same kernels and speed as bf16, forward only): the
    DeiT-B teacher's logits against the reference -- regression bar at ~2x the measured 1.126e-3 (bf16: 6.8e-3; this does
    NOT meet BASELINE.json's 1e-3, see the strict xfail above), top-1 bit-exact; the q/k/v the relation losses read within
    1.5e-3.  The composite and the one-call-per-kernel paths agree bit for bit."""
    from devit_amd import _lib, ops
    s, t, _, _ = models
    g = golden("model_deitb")
    img = torch.from_numpy(det_array("img8", (8, 3, 224, 224))).to(dev)
    try:
        t.precision = "f16"
        t.eval()
        with torch.no_grad():
            logits = t(img)
            d = t(img, output_qkv=True, output_att=True)
            ops.COMPOSITE = False
            logits_granular = t(img)
            ops.COMPOSITE = True
        assert torch.equal(logits, logits_granular)
        e = rel(logits, g["logits"])
        assert chk(e, F16_LOGITS_REGRESSION_BAR), f"f16 teacher logits rel-to-max err {e:.3e}"
        assert np.array_equal(logits.argmax(1).cpu().numpy(), g["top1"])
        q, k, v = d["qkv"][5]
        assert q.dtype == torch.float16 and q.shape == (8, 12, 198, 64) and q.stride() == (198 * 3 * 768, 64, 3 * 768, 1)
        assert chk(rel(q[:2, :, :24], g["q5"]), 1.5e-3) and chk(rel(k[:2, :, :24], g["k5"]), 1.5e-3)
        assert chk(rel(v[:2, :, :24], g["v5"]), 1.5e-3) and chk(rel(d["attention"][5][:2, :24], g["att5"]), 2e-3)
        # forward only: a backward through f16 activations is refused, loudly
        for p in t.parameters():
            p.requires_grad_(True)
        with pytest.raises(_lib.DevitError):
            t(img)
        for p in t.parameters():
            p.requires_grad_(False)
        # the DEKD step with the f16 teacher: teacher logits and the three relation losses move towards the reference
        from devit_amd import engine
        gs = golden("step_bs8")
        dps = torch.from_numpy(gs["dp_scales"]).to(dev)
        s.train()
        with torch.no_grad():
            out = engine.distill_forward(s, t, img, torch.from_numpy(gs["soft_targets"]).to(dev),
                                         dp_scales=[(dps[i, 0].contiguous(), dps[i, 1].contiguous()) for i in range(12)])
        assert chk(rel(out["teacher_logits"], gs["teacher_logits"]), F16_LOGITS_REGRESSION_BAR)   # same quantity as above
        for k_ in ("loss", "cls_loss", "q_loss", "k_loss", "v_loss"):
            assert chk(abs(float(out[k_]) - float(gs[k_])) / abs(float(gs[k_])), 5e-4), k_
    finally:
        ops.COMPOSITE = True
        t.precision = "bf16"
        for p in t.parameters():
            p.requires_grad_(False)
