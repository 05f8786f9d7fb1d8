#!/usr/bin/env python3
"""Would 16-bit MFMA operands with an f16 (11-bit) instead of a bf16 (8-bit) significand bring the BENCHMARKED teacher
forward to BASELINE.json's 1e-3?  (the reference's AMP is fp16, engine.py:32,68; gfx950's f16 MFMA
runs at the bf16 rate; the frozen teacher needs no loss scaling.)

Numerical emulation on the CPU, exact about WHERE the HIP path rounds: the fp32 oracle forward of the DeiT-B teacher
(oracle/devit_oracle.py) with the storage type applied at every point where the bf16 path stores 16-bit values --
GEMM weights, im2row patches, LayerNorm outputs, packed qkv, softmax probabilities P, attention outputs, post-GELU
hidden activations -- fp32 accumulation, fp32 residual stream / LN statistics / softmax statistics / heads, as in the
kernels.  Deviation from the reference goldens (tests/golden/model_deitb.npz), rel = max|a-b| / max|b|.
Writes profiles/r02_f16_emulation.json.  Test/diagnostic tool: imports oracle/, never part of the product path."""
import json, os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import devit_oracle as O
from oracle.detgen import det_array

torch.set_num_threads(8)


def forward(st, geom, img, q, trace=None):
    """trace: optional list that receives the fp32 residual stream after the embedding and after every block."""
    H, depth = geom["num_heads"], geom["depth"]
    w = lambda k: q(st[k])
    B = img.shape[0]
    D = st["patch_embed.proj.weight"].shape[0]
    rows = q(img.reshape(B, 3, 14, 16, 14, 16).permute(0, 2, 4, 1, 3, 5).reshape(B, 196, 768))
    x = rows @ w("patch_embed.proj.weight").reshape(D, 768).t() + st["patch_embed.proj.bias"]
    x = torch.cat([st["cls_token"].expand(B, -1, -1), st["dist_token"].expand(B, -1, -1), x], 1) + st["pos_embed"]
    hd, qkv_mid = D // H, None
    if trace is not None:
        trace.append(x.clone())
    for i in range(depth):
        p = f"blocks.{i}."
        ln1 = q(F.layer_norm(x, (D,), st[p + "norm1.weight"], st[p + "norm1.bias"], O.LN_EPS))
        qkv = q(F.linear(ln1, w(p + "attn.qkv.weight"), st[p + "attn.qkv.bias"]))
        v = qkv.reshape(B, -1, 3, H, hd).permute(2, 0, 3, 1, 4)
        s = (v[0] @ v[1].transpose(-2, -1)) * hd ** -0.5
        m = s.max(-1, keepdim=True).values
        pexp = torch.exp(s - m)
        o = (q(pexp) @ v[2]) / pexp.sum(-1, keepdim=True)          # the kernel rounds the unnormalised P, divides in fp32
        o = q(o.transpose(1, 2).reshape(B, -1, D))
        x = x + F.linear(o, w(p + "attn.proj.weight"), st[p + "attn.proj.bias"])
        ln2 = q(F.layer_norm(x, (D,), st[p + "norm2.weight"], st[p + "norm2.bias"], O.LN_EPS))
        h = q(F.gelu(F.linear(ln2, w(p + "mlp.fc1.weight"), st[p + "mlp.fc1.bias"])))
        x = x + F.linear(h, w(p + "mlp.fc2.weight"), st[p + "mlp.fc2.bias"])
        if i == depth // 2 - 1:
            qkv_mid = v
        if trace is not None:
            trace.append(x.clone())
    x = F.layer_norm(x, (D,), st["norm.weight"], st["norm.bias"], O.LN_EPS)
    lo = F.linear(x[:, 0], st["head.weight"], st["head.bias"])
    lk = F.linear(x[:, 1], st["head_dist.weight"], st["head_dist.bias"])
    return (lo + lk) / 2, qkv_mid


def main():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "model_deitb.npz")))
    geom = O.GEOMETRY["deit_base_distilled_patch16_224"]
    st = O.make_state(geom, 25, "T")
    img = torch.from_numpy(det_array("img8", (8, 3, 224, 224)))
    rel = lambda a, b: float(np.abs(a.numpy() - b).max() / np.abs(b).max())
    out = {}
    with torch.no_grad():
        for name, q in (("f32", lambda t: t), ("bf16", lambda t: t.to(torch.bfloat16).float()),
                        ("f16", lambda t: t.to(torch.float16).float())):
            logits, qkv = forward(st, geom, img, q)
            out[name] = {"logits_rel": rel(logits, g["logits"]), "top1_equal": bool(np.array_equal(logits.argmax(1).numpy(), g["top1"])),
                         "q5_rel": rel(qkv[0][:2, :, :24], g["q5"]), "k5_rel": rel(qkv[1][:2, :, :24], g["k5"]),
                         "v5_rel": rel(qkv[2][:2, :, :24], g["v5"]),
                         "max_abs_16bit_value": float(max(qkv.abs().max(), 1.0))}
            print(name, out[name], flush=True)
    out["note"] = ("CPU emulation of the storage points of the HIP teacher forward (tools/f16_emulation.py); 'bf16' reproduces "
                   "what the kernels measure on MI355X (profiles/r01_parity_report.json: 5.8e-3), which validates the emulation")
    with open(os.path.join(ROOT, "profiles", "r02_f16_emulation.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
