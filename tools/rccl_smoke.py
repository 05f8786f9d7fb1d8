#!/usr/bin/env python3
"""One-GPU smoke test of the RCCL gradient-reduction path: a world-size-1 NCCL(=RCCL) group, with the reducer told it has
two ranks so that every bucket really goes through dist.all_reduce on the side stream while backward is running
(gradients come out halved -- this only checks that the overlap machinery runs, finishes and stays finite)."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
dist.init_process_group("nccl", rank=0, world_size=1)
import devit_amd
from devit_amd import ddp, engine, losses, optim
dev = torch.device("cuda", 0); torch.cuda.set_device(0); B, C = 64, 250
student = devit_amd.create_model("dedeit", num_classes=C, drop_path_rate=0.1, drop_block_rate=None).to(dev).train()
teacher = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=C).to(dev).eval()
for p in teacher.parameters(): p.requires_grad_(False)
flat = ddp.FlatParams(student); ddp.broadcast_parameters(flat); flat.attach_bf16(student)   # broadcast first, then cast the bf16 copies
reducer = ddp.BucketedGradReducer(flat).attach(student); reducer.world = 2
opt = optim.FlatAdamW(flat, lr=1e-4, weight_decay=0.0, max_norm=1.0, ema_decay=0.99996)
crit = losses.DistillLoss(losses.SoftTargetCrossEntropy(), "hard", 0.5, 1.0)
img = torch.randn(B, 3, 224, 224, device=dev); soft = torch.softmax(torch.randn(B, C, device=dev), 1)
look = engine.TeacherLookahead(teacher); look.submit(img)
for it in range(4):
    opt.zero_grad(); t = look.take(img); look.submit(img)
    out = engine.distill_forward(student, teacher, img, soft, criterion=crit, teacher_outputs=t)
    out["loss"].backward(); n = sum(reducer.launched); reducer.finish(); opt.step()
    torch.cuda.synchronize()
    print(f"step {it}: loss {float(out['loss']):.5f}, {len(reducer.buckets)} buckets, {n} all-reduces launched during backward", flush=True)
assert torch.isfinite(flat.flat).all()
dist.destroy_process_group(); print("rccl smoke ok")
