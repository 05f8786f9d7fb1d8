"""Data-parallel gradient exchange for the DEKD step: RCCL all-reduce of reverse-order buckets, overlapped
with backward.  Replaces torch DDP's reducer at distill_sub.py:333.

Design (MI355X: 8 GPUs fully connected by xGMI, ring all-reduce is per-link bound at ~153 GB/s):
  * every parameter's fp32 gradient is a view of ONE flat buffer, laid out in REVERSE forward order, so the
    tensors of a bucket finish together during backward and each bucket is one contiguous slice;
  * the model's autograd nodes report finished parameter groups (`grad_ready` callback: heads, each block,
    embeddings); when the last parameter of a bucket is reported the bucket's all-reduce is launched on a
    side stream (after an event recorded on the compute stream), so it runs under the remaining backward;
  * `finish()` joins the side stream before grad-clip / optimizer.  The buckets hold SUMS; the 1 / world_size of DDP's
    mean is `FlatParams.grad_scale`, which the fused optimizer kernel folds into its clip factor (devit_adamw_step's
    `grad_scale`) -- no extra pass over the 87 MB gradient buffer;
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
        self._w16_mods = []
        # flat_grad holds (true gradient) / grad_scale: 1 normally, 1 / world_size after a summing all-reduce
        # (BucketedGradReducer.finish); consumed by optim.FlatAdamW.step
        self.grad_scale = 1.0

    def attach_bf16(self, model):
        """Keep ONE flat bf16 copy of all parameters (rewritten by the fused optimizer kernel every step) and point
        the GEMM weight caches of the model's Linear / Conv2d children at views of it."""
        if self.flat16 is None:
            self.flat16 = torch.empty(self.numel, dtype=torch.bfloat16, device=self.flat.device)
        self._w16_mods = [mod for mod in model.modules()
                          if isinstance(mod, (torch.nn.Linear, torch.nn.Conv2d)) and getattr(mod, "weight", None) is not None
                          and id(mod.weight) in self.index]
        self.refresh_bf16()
        return self

    def refresh_bf16(self):
        """Re-cast the flat bf16 copy from the fp32 masters and re-stamp the modules' GEMM weight caches.  Needed after
        anything that rewrites the masters behind the parameters' version counters (the initial broadcast writes
        `flat` in place; `p.data = view` keeps each parameter's own counter), and after load_state_dict."""
        if self.flat16 is None:
            return
        from . import ops
        ops.cast_bf16(self.flat, self.flat16)
        for mod in self._w16_mods:
            w = mod.weight
            o = self.offsets[self.index[id(w)]]
            mod._w16 = (w._version, self.flat16[o:o + w.numel()].view(w.shape[0], -1), w.data_ptr())
        self.refresh_kmajor()

    def refresh_kmajor(self):
        """Re-derive the k-major copies (de_vit._w16t: fc2 of a 384-wide model, read by the full-row GEMM) of the weights whose bf16 copies live
        in flat16, after flat16 was rewritten in place (refresh_bf16, optim.FlatAdamW.step): one launch over a cached job table."""
        pairs = []
        for mod in self._w16_mods:
            c, w16 = mod.__dict__.get("_w16t"), mod.__dict__.get("_w16")
            if c is not None and w16 is not None and c[0] == w16[1].data_ptr():
                pairs.append((w16[1], c[2]))
        if not pairs:
            self._kmajor = None
            return
        from . import ops
        key = tuple((a.data_ptr(), b.data_ptr()) for a, b in pairs)
        if getattr(self, "_kmajor", None) is None or self._kmajor.key != key:
            self._kmajor = ops._Transposes(pairs)
        self._kmajor.run()

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


def _report_group(name):
    """Which grad_ready report a parameter belongs to, in backward order: (0, 0) heads and final norm, (1, k) encoder
    block k, (2, 0) embeddings (patch projection, class / distillation tokens, position embedding)."""
    name = name[7:] if name.startswith("module.") else name
    if name.startswith("blocks."):
        return (1, int(name.split(".")[1]))
    if name.startswith(("patch_embed", "pos_embed", "cls_token", "dist_token")):
        return (2, 0)
    return (0, 0)


class _ReadyHook:
    """What BucketedGradReducer.attach() puts on the model as `grad_ready`: the report callback, plus what the encoder's backward asks when it
    groups several blocks' weight gradients into one launch (ops.DeferredWgrads): which bucket a block's parameters travel in, and whether
    there is an exchange to overlap at all."""

    def __init__(self, reducer):
        self.reducer = reducer

    def __call__(self, params):
        self.reducer.mark_ready(params)

    world = property(lambda self: self.reducer.world)

    def bucket_of(self, params):
        for p in params:
            i = self.reducer.flat.index.get(id(p))
            if i is not None:
                return self.reducer.bucket_of[i]
        return None


class BucketedGradReducer:
    """All-reduce (sum) of FlatParams.flat_grad in buckets, fired from `grad_ready` callbacks.

    comm: anything with `.world` and `.all_reduce(flat_f32_view, stream=)` (RcclComm, or a recording stand-in in the
    tests) -> the buckets go through it instead of torch.distributed."""

    def __init__(self, flat: FlatParams, bucket_bytes=25 << 20, process_group=None, comm=None, world=None, plan="layers"):
        self.flat, self.group, self.comm = flat, process_group, comm
        if world is not None:
            self.world = world
        else:
            self.world = comm.world if comm is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        # bucket boundaries on parameter boundaries, in flat (= reverse forward) order.  plan="layers" cuts where backward
        # REPORTS (grad_ready: heads + final norm, then one encoder block at a time, then the embeddings): the first bucket
        # is heads + norm + the last block (it leaves as soon as that block's backward is enqueued), full blocks are
        # grouped up to bucket_bytes, and the embeddings -- whose gradients only exist when backward is over -- travel alone
        # (1.5 MB for `dedeit`: the part of the exchange nothing can overlap).  plan="size": fixed-size cuts.
        groups = [_report_group(n) for n in flat.names]
        self.buckets, start, last = [], 0, 0
        limit = bucket_bytes // 4
        nparams = len(flat.params)
        first_block = next((g for g in groups if g[0] == 1), None)
        final_block = next((g for g in reversed(groups) if g[0] == 1), None)   # block 0: its bucket cannot overlap much, keep it alone
        for i in range(nparams):
            end = flat.offsets[i + 1] if i + 1 < nparams else flat.numel
            last_of_group = i + 1 == nparams or groups[i + 1] != groups[i]
            if plan == "layers" and first_block is not None:
                nxt = groups[i + 1] if i + 1 < nparams else None
                cut = last_of_group and (
                    i + 1 == nparams
                    or groups[i] == first_block                       # heads + norm + last block: out first
                    or nxt[0] == 2                                    # everything before the embeddings
                    or nxt == final_block
                    or (groups[i][0] == 1 and end - start + self._group_elems(groups, i + 1) > limit))
            else:
                cut = end - start >= limit or i + 1 == nparams
            if cut:
                self.buckets.append((start, end, last, i))        # [elem start, elem end), param index range
                start, last = end, i + 1
        self.bucket_of = {}
        for b, (_, _, p0, p1) in enumerate(self.buckets):
            for i in range(p0, p1 + 1):
                self.bucket_of[i] = b
        self.cuda = flat.flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        # CUs the persistent GEMM grids leave free WHILE buckets are in flight (first bucket launched -> finish()).  A
        # collective's workgroups hold CUs for the length of the collective; they cannot share a CU with a GEMM workgroup
        # (LDS and registers are full), so they start when a GEMM ends and a 256-workgroup grid launched meanwhile runs a second,
        # nearly empty round on what is left.  Measured on one GPU (profiles/r03_f_*): a 32-workgroup stand-in kernel starts
        # within 42 us only with 32 CUs reserved (one per shader engine; 160-190 us with 0 / 8 / 16), and keeping 16 / 32 CUs
        # free costs the step 4.3 / 5.2 % whether scoped to this window or permanent (the window is the backward, where
        # the two-workgroups-per-CU GEMMs live).  The exchange itself is ~1 ms of an ~11 ms backward, so even a 2x slowdown
        # of the GEMMs beside it costs less than the reservation: the default is 0.  DEVIT_RESERVE_CUS_EXCHANGE=n turns the
        # scoped form on (what to try first if the 8-GPU curve shows backward stretching under the all-reduce);
        # DEVIT_RESERVE_CUS=n is the library's permanent form.
        import os
        self.reserve_cus = self._parse_reserve(os.environ.get("DEVIT_RESERVE_CUS_EXCHANGE", "0")) if self.cuda else 0
        self._reserved = False
        self._reserved_before = 0        # what devit_get_reserved_cus() said when the window opened (restored by finish())
        self.timing = False          # bench.py: record events around every bucket (allreduce_ms / overlap_frac)
        self.last_timing = None
        self.reset()

    def _group_elems(self, groups, i):
        """Elements of the report group that starts at parameter i."""
        n, j = 0, i
        while j < len(groups) and groups[j] == groups[i]:
            end = self.flat.offsets[j + 1] if j + 1 < len(groups) else self.flat.numel
            n += end - self.flat.offsets[j]
            j += 1
        return n

    def reset(self):
        self.pending = [p1 - p0 + 1 for (_, _, p0, p1) in self.buckets]
        self.reported = set()
        self.handles, self.launched = [], [False] * len(self.buckets)
        self.launch_order = []
        self._events = []

    def attach(self, model):
        """Route the model's grad_ready callbacks here (VisionTransformer.grad_ready)."""
        model.grad_ready = _ReadyHook(self)
        return self

    def mark_ready(self, params):
        for p in params:
            i = self.flat.index.get(id(p))
            if i is None:
                continue
            if i in self.reported:
                raise RuntimeError(f"BucketedGradReducer: parameter {self.flat.names[i]} was reported twice in one backward "
                                   "(its bucket may already be in flight)")
            self.reported.add(i)
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0 and not self.launched[b]:
                self._launch(b)

    def _launch(self, b):
        self.launched[b] = True
        self.launch_order.append(b)
        if self.world == 1:
            return
        s, e, _, _ = self.buckets[b]
        view = self.flat.flat_grad[s:e]
        if self.cuda and self.reserve_cus and not self._reserved:
            from . import _lib as L
            self._reserved_before = L.load().devit_get_reserved_cus()   # a permanent DEVIT_RESERVE_CUS survives the window
            L.call("devit_set_reserved_cus", max(self.reserve_cus, self._reserved_before))      # GEMM grids enqueued from here on leave room for the collective
            self._reserved = True
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())          # backward kernels of this bucket are enqueued
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                if self.comm is not None:
                    self.comm.all_reduce(view, stream=self.stream)
                else:
                    # the collective runs on the process group's own stream, ordered after this side stream; wait() makes
                    # the side stream (not the host) wait for it, so that `finish` only has to join the side stream
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
                if self.timing:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self._events.append((b, e0, e1))
        elif self.comm is not None:
            self.comm.all_reduce(view, stream=None)
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    @staticmethod
    def _parse_reserve(text):
        """DEVIT_RESERVE_CUS_EXCHANGE -> a value devit_set_reserved_cus accepts (a multiple of 8 in [0, 128]), checked HERE:
        a bad value must not surface as an error on the first bucket launch, in the middle of backward, when other ranks may
        already have collectives enqueued."""
        try:
            n = int(text)
        except (TypeError, ValueError):
            raise ValueError(f"DEVIT_RESERVE_CUS_EXCHANGE={text!r}: not an integer")
        if n < 0 or n > 128:
            raise ValueError(f"DEVIT_RESERVE_CUS_EXCHANGE={n}: must lie in [0, 128]")
        return n // 8 * 8

    def finish(self, average=False):
        """Launch whatever was never reported (unused parameters) and join.  The buffer then holds SUMS over ranks and
        `flat.grad_scale` = 1 / world says so (the optimizer kernel applies it); average=True scales in place instead
        (an extra pass; for callers that read `.grad` themselves)."""
        end_bwd = None
        if self.cuda and self.timing and self.world > 1:
            end_bwd = torch.cuda.Event(enable_timing=True)
            end_bwd.record()                                  # backward is enqueued up to here on the compute stream
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        for h in self.handles:
            h.wait()
        if self.cuda and self.world > 1:
            torch.cuda.current_stream().wait_stream(self.stream)
        if self._reserved:
            from . import _lib as L
            L.call("devit_set_reserved_cus", self._reserved_before)
            self._reserved = False
        self.flat.grad_scale = 1.0 / self.world
        if average and self.world > 1:
            self.flat.flat_grad.mul_(1.0 / self.world)
            self.flat.grad_scale = 1.0
        if end_bwd is not None:
            self.last_timing = (self._events, end_bwd)
        order = self.launch_order
        self.reset()
        return order

    def timing_summary(self):
        """(allreduce_ms, overlap_frac) of the last finished step recorded with `timing = True`: the summed bucket
        all-reduce durations (events on the exchange stream) and the share of that time that ran before the last
        backward kernel was done (events on the compute stream).  Synchronises."""
        if self.last_timing is None:
            return None
        events, end_bwd = self.last_timing
        torch.cuda.synchronize()
        base = events[0][1]
        t_bwd = base.elapsed_time(end_bwd)
        total = hidden = 0.0
        for _, e0, e1 in events:
            t0, t1 = base.elapsed_time(e0), base.elapsed_time(e1)
            total += t1 - t0
            hidden += max(0.0, min(t1, t_bwd) - min(t0, t_bwd))
        return total, (hidden / total if total > 0 else 0.0)


def broadcast_parameters(flat: FlatParams, src=0, group=None):
    """DDP's initial parameter broadcast (C4): one collective over the flat master buffer."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat.flat, src=src, group=group)
        flat.refresh_bf16()     # the in-place write does not bump the parameters' version counters (_w16 cannot notice)


def allreduce_mean_(tensors, group=None):
    """In-place mean over ranks of a list of tensors (one coalesced collective): the gradient exchange of the small
    non-flat models (ensemble.py's MultiViT / EnsMLP), DistributedDataParallel's job at ensemble.py:332-334."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / dist.get_world_size(group))
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()


def broadcast_module(module, src=0, group=None):
    """DDP's initial parameter / buffer broadcast for a module that is not flat-packed."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)
    # writing through .data does not bump version counters: drop every cache keyed on them (16-bit GEMM copies, block
    # parameter views), so a forward that ran before the broadcast cannot leave stale weights behind
    for m in module.modules():
        for k in ("_w16", "_w16h", "_w16t", "_bp_cache"):
            m.__dict__.pop(k, None)
