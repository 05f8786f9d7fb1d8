#!/usr/bin/env python3
"""Print measured deviations of both precision modes from the reference goldens (MI355X)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import devit_amd
from devit_amd import engine
from oracle import devit_oracle as O
from oracle.detgen import det_array
dev = torch.device("cuda"); C = 25
G = lambda n: dict(np.load(os.path.join(ROOT, "tests", "golden", n + ".npz")))
gs, gt = O.GEOMETRY["dedeit"], O.GEOMETRY["deit_base_distilled_patch16_224"]
s = devit_amd.create_model("dedeit", num_classes=C, drop_path_rate=0.1); t = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=C)
s.load_state_dict(O.make_state(gs, C, "S")); t.load_state_dict(O.make_state(gt, C, "T")); s.to(dev); t.to(dev).eval()
img = torch.from_numpy(det_array("img8", (8, 3, 224, 224))).to(dev)
rel = lambda a, b: float(np.abs(a.detach().float().cpu().numpy() - b).max() / np.abs(b).max())
out = {}
for prec in ("bf16", "f32"):
    s.precision = t.precision = prec
    r = {}
    for which, m in (("dedeit", s), ("deitb", t)):
        g = G(f"model_{which}"); m.eval()
        with torch.no_grad(): lo = m(img)
        r[f"{which}_logits_rel"] = rel(lo, g["logits"]); r[f"{which}_top1_equal"] = bool(np.array_equal(lo.argmax(1).cpu().numpy(), g["top1"]))
    t.eval(); s.train(); g = G("step_bs8")
    for p in s.parameters(): p.grad = None
    dps = torch.from_numpy(g["dp_scales"]).to(dev)
    o = engine.distill_forward(s, t, img, torch.from_numpy(g["soft_targets"]).to(dev), dp_scales=[(dps[i, 0].contiguous(), dps[i, 1].contiguous()) for i in range(12)])
    for k in ("loss", "cls_loss", "q_loss", "k_loss", "v_loss"): r[k + "_rel"] = abs(float(o[k]) - float(g[k])) / abs(float(g[k]))
    o["loss"].backward()
    names = json.load(open(os.path.join(ROOT, "tests", "golden", "step_param_names.json"))); P = dict(s.named_parameters())
    gn = np.array([P[n].grad.norm().item() for n in names]); r["grad_norm_max_rel"] = float((np.abs(gn - g["grad_norms"]) / g["grad_norms"].max()).max())
    r["g_qkv5_rows_rel"] = rel(P["blocks.5.attn.qkv.weight"].grad[::48], g["g_qkv5_w_rows"])
    out[prec] = r
print(json.dumps(out, indent=1))
