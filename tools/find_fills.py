"""Diagnostic: where do the aten fill kernels of a bench step come from?"""
import os, sys, torch, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import devit_amd
from devit_amd import ddp, engine, losses, optim
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
for _ in range(2): step()
torch.cuda.synchronize()
cnt = collections.Counter()
orig_zero, orig_fill = torch.Tensor.zero_, torch.Tensor.fill_
def wrap(name, fn):
    def f(self, *a, **k):
        st = [l for l in traceback.format_stack(limit=8) if "devit_amd" in l or "bench" in l]
        cnt[(name, str(self.dtype), tuple(self.shape), st[-1].strip().split("\n")[0] if st else "?")] += 1
        return fn(self, *a, **k)
    return f
torch.Tensor.zero_ = wrap("zero_", orig_zero); torch.Tensor.fill_ = wrap("fill_", orig_fill)
for n in ("zeros", "zeros_like", "full", "ones"):
    o = getattr(torch, n)
    def mk(o=o, n=n):
        def f(*a, **k):
            st = [l for l in traceback.format_stack(limit=8) if "devit_amd" in l]
            r = o(*a, **k); cnt[(n, str(r.dtype), tuple(r.shape), st[-1].strip().split("\n")[0] if st else "?")] += 1; return r
        return f
    setattr(torch, n, mk())
step(); torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]): print(v, k)
