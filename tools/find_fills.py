#!/usr/bin/env python3
"""Which torch (non-library) device work does one DEKD step still contain?  aten ops by device time, with shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import devit_amd
from devit_amd import ddp, engine, losses, optim
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda"); B, C = 256, 25
student = devit_amd.create_model("dedeit", num_classes=C, drop_path_rate=0.1, drop_block_rate=None).to(dev).train()
teacher = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=C).to(dev).eval()
for p in teacher.parameters(): p.requires_grad_(False)
flat = ddp.FlatParams(student).attach_bf16(student); reducer = ddp.BucketedGradReducer(flat).attach(student)
opt = optim.FlatAdamW(flat, lr=1e-4, weight_decay=0.0, max_norm=1.0, ema_decay=0.99996)
crit = losses.DistillLoss(losses.SoftTargetCrossEntropy(), "hard", 0.5, 1.0)
img = torch.randn(B, 3, 224, 224, device=dev); soft = torch.softmax(torch.randn(B, C, device=dev), 1)
look = engine.TeacherLookahead(teacher); look.submit(img)
def step():
    opt.zero_grad(); t = look.take(img); look.submit(img)
    out = engine.distill_forward(student, teacher, img, soft, criterion=crit, teacher_outputs=t)
    out["loss"].backward(); reducer.finish(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
import collections
agg = collections.Counter(); tm = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.device_time_total > 0:
        st = [s for s in ev.stack if "devit_amd" in s or "bench" in s or "tools/" in s][:2]
        key = (ev.name, str(ev.input_shapes)[:60], " <- ".join(s.split("/")[-1] for s in st))
        agg[key] += 1; tm[key] += ev.device_time_total
for k, v in sorted(tm.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{v:9.1f} us  x{agg[k]:3d}  {k}")
