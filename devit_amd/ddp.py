"""Data-parallel gradient exchange for the DEKD step: RCCL all-reduce of reverse-order buckets, overlapped
with backward.  Replaces torch DDP's reducer at distill_sub.py:333.

Design (MI355X: 8 GPUs fully connected by xGMI, ring all-reduce is per-link bound at ~153 GB/s):
  * every parameter's fp32 gradient is a view of ONE flat buffer, laid out in REVERSE forward order, so the
    tensors of a bucket finish together during backward and each bucket is one contiguous slice;
  * the model's autograd nodes report finished parameter groups (`grad_ready` callback: heads, each block,
    embeddings); when the last parameter of a bucket is reported the bucket's all-reduce is launched on a
    side stream (after an event recorded on the compute stream), so it runs under the remaining backward;
  * `finish()` joins the side stream before grad-clip / optimizer, and scales by 1 / world_size (mean, like DDP);
  * the teacher is frozen and replicated: no traffic.  87.4 MB fp32 per step at C = 250.
The same code runs on CPU tensors with the gloo backend (tests/test_ddp_gloo.py).
"""
import torch
import torch.distributed as dist


class FlatParams:
    """Re-homes a model's parameters into flat fp32 buffers (master, grad) in reverse registration order.

    `views[name]` are the parameter tensors (nn.Parameter.data now aliases the flat master buffer) and every
    `param.grad` is a view of `flat_grad`, so backward kernels accumulate straight into the bucket memory."""

    def __init__(self, model, pad_to=4):
        params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.names = [n for n, _ in params][::-1]
        ordered = [p for _, p in params][::-1]
        dev, self.params = ordered[0].device, ordered
        self.offsets, off = [], 0
        for p in ordered:
            self.offsets.append(off)
            off += (p.numel() + pad_to - 1) // pad_to * pad_to      # keep every tensor 16-byte aligned
        self.numel = (off + 127) // 128 * 128
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(ordered, self.offsets):
                v = self.flat[o:o + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
        self.index = {id(p): i for i, p in enumerate(ordered)}
        self.flat16 = None

    def attach_bf16(self, model):
        """Keep ONE flat bf16 copy of all parameters (rewritten by the fused optimizer kernel every step) and point
        the GEMM weight caches of the model's Linear / Conv2d children at views of it."""
        from . import ops
        self.flat16 = torch.empty(self.numel, dtype=torch.bfloat16, device=self.flat.device)
        ops.cast_bf16(self.flat, self.flat16)
        for mod in model.modules():
            w = getattr(mod, "weight", None)
            if isinstance(mod, (torch.nn.Linear, torch.nn.Conv2d)) and w is not None and id(w) in self.index:
                o = self.offsets[self.index[id(w)]]
                mod._w16 = (w._version, self.flat16[o:o + w.numel()].view(w.shape[0], -1), w.data_ptr())
        return self

    def zero_grad(self):
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):     # re-attach if someone set grads to None
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)


class RcclComm:
    """The C ABI's own RCCL communicator (devit_comm_*): for hosts that do not route collectives through
    torch.distributed.  The 128-byte rendezvous id travels over whatever process group is already up (any backend; only
    its object broadcast is used) or is handed in by the launcher.  `all_reduce` is asynchronous on the given stream."""

    def __init__(self, rank=None, world=None, unique_id=None, group=None):
        import ctypes as C
        from . import _lib as L
        self._L, self._C = L, C
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if unique_id is None:
            buf = C.create_string_buffer(128)
            if rank == 0:
                L.call("devit_comm_unique_id", buf)
            box = [bytes(buf.raw)]
            if world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            unique_id = box[0]
        self.rank, self.world = rank, world
        self._comm = C.c_void_p()
        L.call("devit_comm_init", C.c_char_p(unique_id), rank, world, C.byref(self._comm))

    def all_reduce(self, flat_f32, stream=None):
        """In-place SUM over ranks of a contiguous fp32 CUDA tensor, enqueued on `stream` (default: current)."""
        L = self._L
        L.require_device(flat_f32)
        if flat_f32.dtype != torch.float32 or not flat_f32.is_contiguous():
            raise L.DevitError("RcclComm.all_reduce: contiguous float32 tensor expected")
        s = stream if stream is not None else torch.cuda.current_stream()
        L.call("devit_comm_allreduce_f32", self._comm, flat_f32.data_ptr(), flat_f32.numel(), s.cuda_stream)

    def destroy(self):
        if self._comm:
            self._L.call("devit_comm_destroy", self._comm)
            self._comm = self._C.c_void_p()


class BucketedGradReducer:
    """All-reduce (mean) of FlatParams.flat_grad in buckets, fired from `grad_ready` callbacks."""

    def __init__(self, flat: FlatParams, bucket_bytes=25 << 20, process_group=None, comm=None):
        """comm: an RcclComm -> the buckets go through the C ABI's communicator instead of torch.distributed's."""
        self.flat, self.group, self.comm = flat, process_group, comm
        self.world = comm.world if comm is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        # bucket boundaries on parameter boundaries, in flat (= reverse forward) order
        self.buckets, start, last = [], 0, 0
        limit = bucket_bytes // 4
        for i, (p, o) in enumerate(zip(flat.params, flat.offsets)):
            end = flat.offsets[i + 1] if i + 1 < len(flat.params) else flat.numel
            if end - start >= limit or i + 1 == len(flat.params):
                self.buckets.append((start, end, last, i))        # [elem start, elem end), param index range
                start, last = end, i + 1
        self.bucket_of = {}
        for b, (_, _, p0, p1) in enumerate(self.buckets):
            for i in range(p0, p1 + 1):
                self.bucket_of[i] = b
        self.cuda = flat.flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.reset()

    def reset(self):
        self.pending = [p1 - p0 + 1 for (_, _, p0, p1) in self.buckets]
        self.handles, self.launched = [], [False] * len(self.buckets)

    def attach(self, model):
        """Route the model's grad_ready callbacks here (VisionTransformer.grad_ready)."""
        model.grad_ready = self.mark_ready
        return self

    def mark_ready(self, params):
        for p in params:
            i = self.flat.index.get(id(p))
            if i is None:
                continue
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0 and not self.launched[b]:
                self._launch(b)

    def _launch(self, b):
        self.launched[b] = True
        if self.world == 1:
            return
        s, e, _, _ = self.buckets[b]
        view = self.flat.flat_grad[s:e]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())          # backward kernels of this bucket are enqueued
            self.stream.wait_event(ev)
            if self.comm is not None:
                self.comm.all_reduce(view, stream=self.stream)
                return
            with torch.cuda.stream(self.stream):
                self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Launch whatever was never reported (unused parameters), wait, and turn sums into means."""
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        for h in self.handles:
            h.wait()
        if self.cuda and self.world > 1:
            torch.cuda.current_stream().wait_stream(self.stream)
        if self.world > 1:
            self.flat.flat_grad.mul_(1.0 / self.world)
        self.reset()


def broadcast_parameters(flat: FlatParams, src=0, group=None):
    """DDP's initial parameter broadcast (C4): one collective over the flat master buffer."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat.flat, src=src, group=group)
