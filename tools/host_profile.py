#!/usr/bin/env python3
"""cProfile of the host side of the DEKD step (enqueue only; the device runs behind)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import devit_amd
from devit_amd import ddp, engine, losses, optim
dev = torch.device("cuda"); B, C = 256, 25
student = devit_amd.create_model("dedeit", num_classes=C, drop_path_rate=0.1, drop_block_rate=None).to(dev).train()
teacher = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=C).to(dev).eval()
for p in teacher.parameters(): p.requires_grad_(False)
flat = ddp.FlatParams(student); flat.attach_bf16(student); reducer = ddp.BucketedGradReducer(flat).attach(student)
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
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
