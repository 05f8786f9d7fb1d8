"""world_size-2 gloo test of the data-parallel gradient exchange (devit_amd/ddp.py) on CPU: bucketed, reverse-order
all-reduce fired from grad_ready callbacks must equal the mean of the per-rank gradients."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from devit_amd import ddp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(7)                       # same init on every rank
    model = torch.nn.Sequential(torch.nn.Linear(24, 48), torch.nn.GELU(), torch.nn.Linear(48, 48), torch.nn.Linear(48, 5))
    if rank == 1:                              # rank 1 starts different: the initial broadcast must fix it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    flat = ddp.FlatParams(model)
    # the bf16 GEMM copies are cast by the HIP library on the GPU; here a torch cast stands in so that the ORDER
    # (broadcast, then re-cast) is what gets tested: attach before the broadcast, as a careless caller would
    from devit_amd import ops
    ops.cast_bf16 = lambda src, dst=None: dst.copy_(src)
    flat.attach_bf16(model)
    ddp.broadcast_parameters(flat)
    bf16_ok = torch.equal(flat.flat16, flat.flat.to(torch.bfloat16)) and all(
        torch.equal(m._w16[1].reshape(-1), m.weight.detach().to(torch.bfloat16).reshape(-1)) for m in model if isinstance(m, torch.nn.Linear))
    red = ddp.BucketedGradReducer(flat, bucket_bytes=4096)
    assert len(red.buckets) >= 2
    torch.manual_seed(100 + rank)              # different data per rank
    x, y = torch.randn(16, 24), torch.randn(16, 5)
    ref = torch.nn.Sequential(torch.nn.Linear(24, 48), torch.nn.GELU(), torch.nn.Linear(48, 48), torch.nn.Linear(48, 5))
    ref.load_state_dict(model.state_dict())
    (ref(x) - y).square().mean().backward()
    local = [p.grad.clone() for p in ref.parameters()]
    # emulate the HIP path: kernels accumulate into param.grad, groups are reported in backward order
    flat.zero_grad()
    params = list(model.parameters())
    for p, g in reversed(list(zip(params, local))):
        p.grad.add_(g)
        red.mark_ready([p])
    order = red.finish()
    assert order == sorted(order) and len(order) == len(red.buckets)     # buckets leave in flat (= reverse forward) order
    assert flat.grad_scale == 1.0 / world                                  # sums in the buffer, the mean in the scale
    gathered = [torch.zeros_like(torch.cat([g.reshape(-1) for g in local])) for _ in range(world)]
    dist.all_gather(gathered, torch.cat([g.reshape(-1) for g in local]))
    mean = sum(gathered) / world
    got = torch.cat([p.grad.reshape(-1) for p in params]) * flat.grad_scale
    ok = torch.allclose(got, mean, rtol=1e-6, atol=1e-7) and bf16_ok
    # a parameter reported twice in one backward is a bug in the autograd nodes: refuse it
    red.mark_ready([params[0]])
    try:
        red.mark_ready([params[0]])
        ok = False
    except RuntimeError:
        red.reset()
    w0 = torch.cat([p.detach().reshape(-1) for p in params])
    allw = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(allw, w0)
    ok = ok and torch.equal(allw[0], allw[1])
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def _module_sync_worker(rank, world, port, q):
    """The ensemble stage's synchronisation (ensemble.py:332-334 wraps both models in DistributedDataParallel; here
    ddp.broadcast_module + ddp.allreduce_mean_): ranks that start different must end with rank 0's parameters and
    buffers, and with the rank-mean of their gradients -- a None gradient (a frozen or unused parameter) is skipped."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)              # different init per rank, as the CLIs seed with seed + rank
    model = torch.nn.Sequential(torch.nn.Linear(6, 10), torch.nn.BatchNorm1d(10), torch.nn.Linear(10, 3))
    ddp.broadcast_module(model)
    state = torch.cat([t.detach().reshape(-1).float() for t in list(model.parameters()) + list(model.buffers())])
    both = [torch.zeros_like(state) for _ in range(world)]
    dist.all_gather(both, state)
    ok = torch.equal(both[0], both[1])
    params = list(model.parameters())
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))      # rank-dependent: the mean is 1.5 (i + 1)
    params[-1].grad = None                                         # skipped on every rank, must stay None
    ddp.allreduce_mean_([p.grad for p in params])
    ok = ok and params[-1].grad is None
    for i, p in enumerate(params[:-1]):
        ok = ok and torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1)), rtol=1e-6, atol=0)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_module_broadcast_and_gradient_mean_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_module_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_layer_bucket_plan():
    """plan="layers" (default): buckets are cut where backward reports -- heads + final norm + the last block leave first, whole
    blocks are grouped up to bucket_bytes, block 0 and the embeddings travel alone so that what no backward kernel can overlap
    stays small (reverse-layer-order buckets launched as each bucket's gradients complete)."""
    import devit_amd
    m = devit_amd.create_model("dedeit", num_classes=250)
    flat = ddp.FlatParams(m)
    red = ddp.BucketedGradReducer(flat, world=2)
    names = flat.names
    spans = [(names[p0], names[p1], (e - s) * 4) for s, e, p0, p1 in red.buckets]
    # contiguous cover of the flat buffer, in flat (= reverse forward) order
    assert red.buckets[0][0] == 0 and red.buckets[-1][1] == flat.numel
    assert all(a[1] == b[0] and a[3] + 1 == b[2] for a, b in zip(red.buckets, red.buckets[1:]))
    first = names[red.buckets[0][2]:red.buckets[0][3] + 1]
    assert first[0].startswith("head") and any(n.startswith("norm.") for n in first) and first[-1].startswith("blocks.11.")
    assert not any(n.startswith("blocks.10.") for n in first)
    last = names[red.buckets[-1][2]:red.buckets[-1][3] + 1]
    assert all(n.startswith(("patch_embed", "pos_embed", "cls_token", "dist_token")) for n in last) and spans[-1][2] <= 3 << 20
    b0 = names[red.buckets[-2][2]:red.buckets[-2][3] + 1]
    assert all(n.startswith("blocks.0.") for n in b0)
    assert all(sz <= (25 << 20) for _, _, sz in spans)
    # every block sits in exactly one bucket
    for k in range(12):
        owners = {red.bucket_of[i] for i, n in enumerate(names) if n.startswith(f"blocks.{k}.")}
        assert len(owners) == 1
    # a model without encoder blocks falls back to fixed-size cuts
    t = torch.nn.Sequential(torch.nn.Linear(24, 48), torch.nn.Linear(48, 48), torch.nn.Linear(48, 5))
    assert len(ddp.BucketedGradReducer(ddp.FlatParams(t), bucket_bytes=4096, world=2).buckets) >= 2
