"""Pin the CPU oracle against the golden vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only; same ATen ops in (nearly) the same order, so the
tolerances are a few fp32 ulps of the value range, not a precision claim."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import devit_oracle as O
from oracle.detgen import det_array, det_labels

C = 25
GS, GT = O.GEOMETRY["dedeit"], O.GEOMETRY["deit_base_distilled_patch16_224"]


def close(a, b, rtol=2e-5, atol=2e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(a - b).max())
    assert err <= atol + rtol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.fixture(scope="module")
def states():
    return O.make_state(GS, C, "S"), O.make_state(GT, C, "T")


@pytest.fixture(scope="module")
def img():
    return torch.from_numpy(det_array("img8", (8, 3, 224, 224)))


def test_statedict_contract(states):
    with open(os.path.join(os.path.dirname(__file__), "golden", "statedict_keys.json")) as f:
        c = json.load(f)
    assert [[k, list(v.shape)] for k, v in states[0].items()] == c["dedeit_keys"]
    assert [[k, list(v.shape)] for k, v in states[1].items()] == c["deitb_keys"]
    assert len(c["dedeit_keys"]) == 155
    assert [k for k, _ in c["dedeit_keys"]][-4:] == ["head.weight", "head.bias", "head_dist.weight", "head_dist.bias"]
    assert sum(v.numel() for v in states[0].values()) == c["n_params_dedeit_c25"] == 21685682
    assert sum(v.numel() for v in states[1].values()) == c["n_params_deitb_c25"] == 85838642
    assert c["n_mlp"] == 12 and c["n_attn"] == 12


@pytest.mark.parametrize("tag,geom,D,H", [("S", GS, 384, 6), ("T", GT, 768, 12)])
def test_modules(golden, states, tag, geom, D, H):
    g = golden(f"module_{tag}")
    st = states[0] if tag == "S" else states[1]
    x = torch.from_numpy(det_array(f"x/{tag}", (2, 198, D))).requires_grad_(True)
    sub = lambda t: t[:, ::9]
    y, _ = O.mlp(st, "blocks.5.mlp.", x)
    close(sub(y), g["mlp_y"])
    w1 = st["blocks.5.mlp.fc1.weight"].requires_grad_(True)
    y, _ = O.mlp(st, "blocks.5.mlp.", x)
    gx, gw = torch.autograd.grad(y.square().sum(), [x, w1])
    close(sub(gx), g["mlp_dx"], rtol=1e-4)
    close(gw[:8], g["mlp_dw1_rows"], rtol=1e-4)
    st["blocks.5.mlp.fc1.weight"] = w1.detach()
    ng = torch.from_numpy(g["neuron_gate"])
    y2, h = O.mlp(st, "blocks.5.mlp.", x, ng)
    close(sub(y2), g["mlp_y_gated"])
    close(h.sum(dim=(0, 1)), g["mlp_neuron_output_sum"], rtol=1e-4)   # post-mask values (Q3)
    a, (q, k, v), _ = O.attention(st, "blocks.5.attn.", x, H)
    close(sub(a), g["attn_y"])
    close(q[:, :, :16], g["attn_q"]); close(k[:, :, :16], g["attn_k"]); close(v[:, :, :16], g["attn_v"])
    ga, = torch.autograd.grad(a.square().sum(), [x])
    close(sub(ga), g["attn_dx"], rtol=1e-4)
    hg = torch.from_numpy(g["head_gate"])
    a2, _, ho = O.attention(st, "blocks.5.attn.", x, H, hg)
    close(sub(a2), g["attn_y_gated"])
    close(ho.sum(dim=(0, 1)), g["attn_head_output_sum"], rtol=1e-4)
    bx, _, att = O.block(st, 5, x, H)
    close(sub(bx), g["block_y"]); close(sub(att), g["block_att"])


@pytest.mark.parametrize("tag,geom,si", [("dedeit", GS, 0), ("deitb", GT, 1)])
def test_model(golden, states, img, tag, geom, si):
    g = golden(f"model_{tag}")
    with torch.no_grad():
        o = O.forward(states[si], geom, img, training=False)
        tr = O.forward(states[si], geom, img, training=True)
    close(o["output"], g["logits"], rtol=5e-5)
    assert np.array_equal(o["output"].argmax(1).numpy(), g["top1"])       # top-1 bit-exact
    close(tr["output"][0], g["train_cls"], rtol=5e-5); close(tr["output"][1], g["train_dist"], rtol=5e-5)
    q, k, v = o["qkv"][5]
    close(q[:2, :, :24], g["q5"], rtol=5e-5); close(k[:2, :, :24], g["k5"], rtol=5e-5)
    close(v[:2, :, :24], g["v5"], rtol=5e-5)
    close(o["attention"][5][:2, :24], g["att5"], rtol=5e-5)
    close(o["encoder"][-1][:2, :24], g["enc_last"], rtol=5e-5)
    close(o["last_tokens"][0], g["last_cls"], rtol=5e-5); close(o["last_tokens"][1], g["last_dist"], rtol=5e-5)
    es = np.stack([[e.mean().item(), e.abs().mean().item()] for e in o["encoder"]])
    close(es, g["enc_stats"], rtol=1e-4)
    # bf16 autocast run of the reference: documents what a bf16 pipeline can reach
    gb = golden(f"model_{tag}_bf16")["logits"]
    rel = np.abs(gb - g["logits"]).max() / np.abs(g["logits"]).max()
    assert rel < 5e-2 and np.array_equal(gb.argmax(1), g["top1"])


def test_cls_loss(golden):
    g = golden("loss_cls")
    lo = torch.from_numpy(det_array("lo", (8, C), std=1.5)).requires_grad_(True)
    lk = torch.from_numpy(det_array("lk", (8, C), std=1.5)).requires_grad_(True)
    lt = torch.from_numpy(det_array("lt", (8, C), std=2.0))
    soft = torch.from_numpy(g["soft_targets"])
    for kind, tau in (("hard", 1.0), ("soft", 3.0)):
        l = O.distill_cls_loss(lo, lk, lt, soft, kind, 0.5, tau)
        d = torch.autograd.grad(l, [lo, lk])
        close(l, g[f"{kind}_loss"]); close(d[0], g[f"{kind}_dlo"]); close(d[1], g[f"{kind}_dlk"])


def test_cls_loss_hard_labels(golden):
    """DistillLoss with the base criteria of distill_sub.py:345-352 for mixup off: LabelSmoothingCrossEntropy(0.1), CrossEntropyLoss."""
    g = golden("loss_cls_hardlabels")
    lo = torch.from_numpy(det_array("lo", (8, C), std=1.5)).requires_grad_(True)
    lk = torch.from_numpy(det_array("lk", (8, C), std=1.5)).requires_grad_(True)
    lt = torch.from_numpy(det_array("lt", (8, C), std=2.0))
    y = torch.from_numpy(g["labels"])
    for base in ("ls", "ce"):
        for kind, tau in (("none", 1.0), ("hard", 1.0), ("soft", 3.0)):
            l = O.distill_cls_loss(lo, lk, lt, y, kind, 0.5, tau, base=base, smoothing=0.1)
            d = torch.autograd.grad(l, [lo, lk], allow_unused=True)
            close(l, g[f"{base}_{kind}_loss"]); close(d[0], g[f"{base}_{kind}_dlo"])
            close(d[1] if d[1] is not None else torch.zeros_like(lk), g[f"{base}_{kind}_dlk"])


def test_relation_loss(golden):
    g = golden("loss_relation")
    tf = torch.from_numpy(det_array("tf", (2, 198, 3, 12, 64), std=0.25)).permute(2, 0, 3, 1, 4)[1]
    sf = torch.from_numpy(det_array("sf", (2, 198, 3, 6, 64), std=0.25)).permute(2, 0, 3, 1, 4)[1]
    sf = sf.detach().requires_grad_(True)
    l = O.feature_relation_loss(tf, sf)
    d, = torch.autograd.grad(l, [sf])
    assert float(g["loss"]) > 1e-3     # fixture is not in the degenerate one-hot regime
    close(l, g["loss"], rtol=1e-5); close(d[:, :, ::9], g["dstudent"], rtol=1e-4)


def test_distill_step(golden, states, img):
    g = golden("step_bs8")
    st_s = {k: v.clone().requires_grad_(True) for k, v in states[0].items()}
    dps = torch.from_numpy(g["dp_scales"])
    out = O.distill_step(st_s, GS, states[1], GT, img, torch.from_numpy(g["soft_targets"]),
                         dp_scales=[(dps[i, 0], dps[i, 1]) for i in range(12)])
    for k in ("loss", "cls_loss", "q_loss", "k_loss", "v_loss"):
        close(out[k], g[k], rtol=2e-5)
    close(out["teacher"]["output"], g["teacher_logits"], rtol=5e-5)
    out["loss"].backward()
    with open(os.path.join(os.path.dirname(__file__), "golden", "step_param_names.json")) as f:
        names = json.load(f)
    gn = np.array([st_s[n].grad.norm().item() for n in names])
    assert np.abs(gn - g["grad_norms"]).max() <= 2e-4 * g["grad_norms"].max()
    close(st_s["head.weight"].grad, g["g_head_w"], rtol=1e-4)
    close(st_s["blocks.5.attn.qkv.weight"].grad[::48], g["g_qkv5_w_rows"], rtol=1e-4)
    close(st_s["blocks.0.mlp.fc1.weight"].grad[::64], g["g_fc1_0_rows"], rtol=1e-4)
    close(st_s["pos_embed"].grad[0, ::8], g["g_pos"], rtol=1e-4)
    close(st_s["patch_embed.proj.weight"].grad[::16].reshape(-1, 768), g["g_patch_w"], rtol=1e-4)


def test_ensemble_stage(golden):
    """MultiViT + EnsMLP + EnsLoss restatement vs the reference's own modules (config 5 path)."""
    g = golden("ensemble")
    subs = [O.make_state(GS, 25, f"E{i}") for i in range(4)]
    ens = {k: v.clone().requires_grad_(True) for k, v in O.make_ens_state().items()}
    img = torch.from_numpy(det_array("img4", (4, 3, 224, 224)))
    with torch.no_grad():
        _, le = O.ens_forward(subs, GS, ens, img, training=False)
    close(le, g["logits_eval"], rtol=5e-5)
    tokens, logits = O.ens_forward(subs, GS, ens, img, training=True)
    close(logits, g["logits_train"], rtol=5e-5); close(tokens[0], g["tok_cls"], rtol=5e-5); close(tokens[1], g["tok_dist"], rtol=5e-5)
    with torch.no_grad():
        tea = O.forward(O.make_state(GT, 100, "T100"), GT, img, training=False)
    tl, cl = O.ens_loss(tokens, logits, tea, torch.from_numpy(g["soft_targets"]))
    close(tl, g["token_loss"], rtol=2e-5); close(cl, g["cls_loss"], rtol=2e-5)
    (tl + cl).backward()
    close(ens["cls_mlp.weight"].grad[::48], g["g_cls_mlp_w_rows"], rtol=1e-4)
    close(ens["dist_classifier.weight"].grad[::10], g["g_dist_cls_w"], rtol=1e-4)
    assert int(g["n_multi_keys"]) == 604
