"""The data-parallel exchange on the REAL student (config 4, distill_sub.py:332-334) as far as one GPU allows: a
recording communicator stands in for RCCL with the world size forced to 2.  Checked: every one of the 155 parameters
is reported exactly once per backward by the autograd nodes (HeadsFn / EncoderFn / PatchEmbedFn), every bucket's
all-reduce is launched DURING backward (before finish()) and in flat = reverse-layer order, the exchanged buffer is
the sum over the (two identical) ranks, and the optimizer's grad_scale turns it back into the mean."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class RecordingComm:
    """world = 2, every rank holding the same gradient: the all-reduce (SUM) doubles the bucket in place, on the stream it
    is given -- like RCCL, asynchronously."""

    def __init__(self):
        self.world, self.calls, self.in_backward = 2, [], False

    def all_reduce(self, view, stream=None):
        self.calls.append((view.data_ptr(), view.numel(), self.in_backward))
        with torch.cuda.stream(stream):
            view.mul_(2.0)


def _step(student, teacher, img, soft):
    from devit_amd import engine
    out = engine.distill_forward(student, teacher, img, soft, dp_scales=None)
    out["loss"].backward()
    return out


def test_real_model_reports_every_parameter_once_and_buckets_overlap():
    import devit_amd
    from devit_amd import ddp, optim
    dev = torch.device("cuda")
    torch.manual_seed(0)
    C, B = 25, 4
    student = devit_amd.create_model("dedeit", num_classes=C, drop_path_rate=0.0).to(dev).train()
    teacher = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=C).to(dev).eval()
    for p in teacher.parameters():
        p.requires_grad_(False)
    img = torch.randn(B, 3, 224, 224, device=dev)
    soft = torch.full((B, C), 0.1 / C, device=dev)
    soft[:, 3] += 0.9

    flat = ddp.FlatParams(student)
    flat.attach_bf16(student)
    assert len(flat.params) == 155

    # reference: the local gradient, no exchange
    flat.zero_grad()
    _step(student, teacher, img, soft)
    torch.cuda.synchronize()
    local = flat.flat_grad.clone()
    assert float(local.abs().max()) > 0

    comm = RecordingComm()
    red = ddp.BucketedGradReducer(flat, bucket_bytes=8 << 20, comm=comm).attach(student)   # 87 MB -> ~11 buckets
    assert red.world == 2 and len(red.buckets) >= 8
    seen = []
    inner = red.mark_ready
    student.grad_ready = lambda params: (seen.extend(flat.index[id(p)] for p in params), inner(params))[1]

    flat.zero_grad()
    comm.in_backward = True
    _step(student, teacher, img, soft)
    comm.in_backward = False
    launched_in_backward = len(comm.calls)
    order = red.finish()
    torch.cuda.synchronize()

    assert sorted(seen) == list(range(155)), "every parameter exactly once"            # (a double report raises already)
    assert launched_in_backward == len(red.buckets) == len(comm.calls), "all buckets must leave during backward"
    assert all(flag for _, _, flag in comm.calls)
    assert order == list(range(len(red.buckets))), "reverse-layer (= flat) order"
    base = flat.flat_grad.data_ptr()
    assert [(p - base) // 4 for p, _, _ in comm.calls] == [s for s, _, _, _ in red.buckets]
    assert sum(n for _, n, _ in comm.calls) == flat.numel
    # heads' bucket first, the patch embedding's last
    assert flat.names[0].startswith("head") and flat.names[-1] in ("cls_token", "pos_embed", "dist_token")
    assert flat.grad_scale == 0.5
    # same kernels in the same order; the split-K weight gradients leave through fp32 atomics, so up to summation order
    torch.testing.assert_close(flat.flat_grad, 2.0 * local, rtol=1e-4, atol=1e-5 * float(local.abs().max()))

    # the optimizer kernel folds 1 / world back in (devit_adamw_step's grad_scale): after one step from zero moments the
    # first moment is (1 - beta1) * clip * mean gradient, with the clip factor computed from the MEAN gradient's norm
    opt = optim.FlatAdamW(flat, lr=1e-3, max_norm=1.0)
    opt.step()
    torch.cuda.synchronize()
    assert flat.grad_scale == 1.0                                  # consumed
    nrm = float(local.double().norm())
    clip = min(1.0, 1.0 / (nrm + 1e-6))
    torch.testing.assert_close(opt.m, 0.1 * clip * local, rtol=1e-4, atol=1e-5 * clip * float(local.abs().max()))
    torch.testing.assert_close(float(opt.gnorm_sq) ** 0.5, 2.0 * nrm, rtol=1e-4, atol=0)   # the raw buffer held the sum


def test_weight_decay_groups_match_torch_adamw():
    """timm create_optimizer (distill_sub.py:340): no decay for 1-D tensors, biases and no_weight_decay() names."""
    import devit_amd
    from devit_amd import ddp, optim
    dev = torch.device("cuda")
    torch.manual_seed(1)
    m = devit_amd.create_model("dedeit", num_classes=10).to(dev)
    skip = optim.no_decay_names(m)
    assert {"pos_embed", "cls_token", "dist_token", "blocks.0.norm1.weight", "blocks.3.mlp.fc1.bias", "head.bias"} <= skip
    assert "blocks.0.attn.qkv.weight" not in skip and "patch_embed.proj.weight" not in skip
    ref = {n: p.detach().clone() for n, p in m.named_parameters()}
    flat = ddp.FlatParams(m)
    g = torch.randn_like(flat.flat) * 1e-2
    flat.flat_grad.copy_(g)
    opt = optim.FlatAdamW(flat, lr=1e-2, weight_decay=0.05, no_decay=skip)
    opt.step()
    torch.cuda.synchronize()
    params = {n: torch.nn.Parameter(v.clone()) for n, v in ref.items()}
    topt = torch.optim.AdamW([{"params": [p for n, p in params.items() if n not in skip], "weight_decay": 0.05},
                              {"params": [p for n, p in params.items() if n in skip], "weight_decay": 0.0}], lr=1e-2)
    for name, p, o in zip(flat.names, flat.params, flat.offsets):
        params[name].grad = g[o:o + p.numel()].view_as(p).clone()
    topt.step()
    for n, p in m.named_parameters():
        d = float((p.detach() - params[n].detach()).abs().max())
        assert d < 1e-6, (n, d, float((p.detach() - ref[n]).abs().max()))


def test_two_ranks_share_one_gpu():
    """A world-size-2 run of the real DEKD step on hardware, as far as one GPU allows (BASELINE config 4: dedeit <- DeiT-B,
    C = 250, DDP).  Two ranks, both on cuda:0, different initial weights and different data; exchange through gloo (RCCL
    refuses two ranks on one device).  The worker (tests/_ddp_two_ranks_worker.py) asserts identical masters on both ranks
    after two steps, the mean gradient against one process on the concatenated batch, every bucket launched during
    backward, and the reducer's timing summary; this test checks it ran to the end on both ranks."""
    import json
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "tests", "_ddp_two_ranks_worker.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("DDP_REHEARSAL ")]
    assert len(lines) == 1, r.stdout[-1500:]
    res = json.loads(lines[0][len("DDP_REHEARSAL "):])
    assert res["world"] == 2 and res["params"] == 155 and res["buckets"] >= 4
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "ddp_two_ranks_one_gpu.json"), "w") as f:
        json.dump(res, f, indent=1)


def test_exchange_through_torch_distributed_nccl_at_world_size_one():
    """The DEFAULT transport at N > 1 -- torch.distributed's NCCL (= RCCL) process group, all_reduce(async_op=True) on the
    exchange stream per bucket during backward -- exercised by a `-m gpu` test as far as one GPU allows (what
    `bench.py --rehearse-exchange torch` does): a world-size-1 NCCL group, the reducer told it has two ranks.  Two real steps
    (clip 1.0, AdamW, EMA) must leave the masters where a run WITHOUT the exchange leaves them when its optimizer is handed the
    same 1 / world: RCCL at one rank moves no data, so any difference would be the exchange machinery (stream joins, bucket views,
    grad_scale)."""
    import os
    import socket
    import torch.distributed as dist
    import devit_amd
    from devit_amd import ddp, optim
    dev = torch.device("cuda")
    C, B = 250, 4                                         # config 4's class count
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.manual_seed(11)
    img = torch.randn(B, 3, 224, 224, device=dev)
    soft = torch.full((B, C), 0.1 / C, device=dev)
    soft[:, 7] += 0.9
    teacher = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=C).to(dev).eval()
    for p in teacher.parameters():
        p.requires_grad_(False)

    def run(exchange):
        torch.manual_seed(5)
        student = devit_amd.create_model("dedeit", num_classes=C, drop_path_rate=0.0).to(dev).train()
        flat = ddp.FlatParams(student)
        flat.attach_bf16(student)
        init = flat.flat.clone()
        red = ddp.BucketedGradReducer(flat, world=2 if exchange else None).attach(student)
        opt = optim.FlatAdamW(flat, lr=1e-3, max_norm=1.0, ema_decay=0.99996)
        orders, first_grad = [], None
        for k in range(2):
            opt.zero_grad()
            _step(student, teacher, img, soft)
            orders.append(red.finish())
            if k == 0:
                torch.cuda.synchronize()
                first_grad = flat.flat_grad.clone()      # what the exchange left in the buffer (a one-rank all-reduce is the identity)
            if exchange:
                assert flat.grad_scale == 0.5
            else:
                flat.grad_scale = 0.5                     # the same 1 / world, without the exchange
            opt.step()
        torch.cuda.synchronize()
        return flat.flat.clone(), init, first_grad, red, orders

    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        m1, init, g1, red, orders = run(True)
        assert red.world == 2 and len(red.buckets) >= 4 and red.comm is None       # torch.distributed transport, not the C-ABI one
        assert all(o == list(range(len(red.buckets))) for o in orders)              # every bucket launched, in flat order
    finally:
        dist.destroy_process_group()
    m0, init0, g0, _, _ = run(False)
    assert torch.equal(init, init0)
    # the buffer after the exchange is the local gradient (same kernels, same inputs; the split-K weight gradients leave through
    # fp32 atomics, so up to summation order)
    torch.testing.assert_close(g1, g0, rtol=1e-4, atol=1e-5 * float(g0.abs().max()))
    # two AdamW steps later: AdamW's first steps move every element by ~lr * sign(g), so an element whose gradient is inside the
    # atomics' summation noise may go the other way (the same between any two runs of this step); the MOVEMENT must agree
    mov1, mov0 = m1 - init, m0 - init
    rel = float((mov1 - mov0).norm() / mov0.norm())
    assert float(mov0.abs().max()) > 1e-3 and rel < 2e-2, rel
