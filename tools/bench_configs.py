#!/usr/bin/env python3
"""Secondary numbers of BASELINE.md §4 (not the headline): config 2 (DeiT-B teacher eval forward, bs 256) and config 5
(4 x shrunk dedeit + EnsMLP collaborative inference, C = 1000, bs 256) on one MI355X.  One JSON line per config."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import devit_amd
from devit_amd.ensemble_models import EnsMLP, MultiViT
dev = torch.device("cuda"); B = 256
img = torch.randn(B, 3, 224, 224, device=dev)
def timed(fn, steps=20, warmup=5):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps
with torch.no_grad():
    teacher = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=25).to(dev).eval()
    t = timed(lambda: teacher(img))
    print(json.dumps({"config": "2: deit_base_distilled_patch16_224 eval forward bf16 bs256 (public forward: every row of every block)", "images_per_sec": round(B / t, 1),
                      "ms": round(t * 1e3, 3), "frac_of_2.5PF": round(B / t * 35.311e9 / 2.5e15, 4)}))
    from devit_amd import de_vit

    def lean():
        with de_vit.lean_tail(teacher):          # what engine.evaluate / the DEKD step run: the last block on its token rows
            return teacher(img)
    t = timed(lean)
    print(json.dumps({"config": "2: the same inside de_vit.lean_tail (engine.evaluate, DEKD teacher): logits bit-identical", "images_per_sec": round(B / t, 1),
                      "ms": round(t * 1e3, 3), "frac_of_2.5PF_algorithmic": round(B / t * 35.311e9 / 2.5e15, 4)}))
    del teacher
    multi = MultiViT("dedeit", drop=0, drop_path=0.0, num_classes_list=[250] * 4, num_div=4).to(dev).eval()
    ens = EnsMLP("dedeit", 1000, 384, [250] * 4, 768).to(dev).eval()
    g = torch.Generator().manual_seed(0)
    for bb in multi.backbones:            # shrink_ratio 0.3: 0/1 gates on heads and neurons (masking)
        for blk in bb.blocks:
            hm = torch.ones(6); hm[torch.randperm(6, generator=g)[:2]] = 0
            nm = torch.ones(1536); nm[torch.randperm(1536, generator=g)[:461]] = 0
            blk.attn.gate, blk.mlp.gate = hm, nm
    t = timed(lambda: ens(multi(img)))
    print(json.dumps({"config": "5: 4 x dedeit (30% head/neuron gates) + EnsMLP inference, C=1000, bs256", "images_per_sec": round(B / t, 1),
                      "ms": round(t * 1e3, 3), "frac_of_2.5PF_dense": round(B / t * 4 * 9.247e9 / 2.5e15, 4)}))
    from devit_amd import shrink
    shrink.compact(multi)                 # physical shrinking: same function, smaller GEMMs
    gf = shrink.compacted_gflops(multi, num_classes=250)
    t = timed(lambda: ens(multi(img)))
    print(json.dumps({"config": "5c: the same 4 sub-models physically shrunk (shrink.compact) + EnsMLP", "images_per_sec": round(B / t, 1),
                      "ms": round(t * 1e3, 3), "gflop_per_img_run": round(gf, 3), "frac_of_2.5PF_run": round(B / t * gf * 1e9 / 2.5e15, 4)}))
