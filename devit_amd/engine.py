"""Step loops with the reference's names and semantics (engine.py), hosted on the HIP path.

  distill_forward   : engine.py:68-106 (student train forward, teacher eval forward, DEKD losses)
  train_1epoch_qkv  : engine.py:48-140
  evaluate          : engine.py:17-45
The five `.item()` host syncs + `cuda.synchronize()` per step of the reference (engine.py:108-117,130) are
kept for logging parity in train_1epoch_qkv, but only every `print_freq` steps.
"""
import math
import os
import sys

import torch

from . import de_vit, losses, ops


def _hand_over(out, main):
    """Outputs of a side-stream forward that the main stream will read.  Small tensors from the side stream's allocator
    pool are marked with record_stream(); the packed qkv buffers of the composite path are views of an arena that was
    allocated from the MAIN stream's pool (ops.ARENA_ALLOC_STREAM) and need no mark -- see the note there."""
    o = out['output']
    for t in (o if isinstance(o, tuple) else (o,)):
        t.record_stream(main)
    for qkv in out['qkv']:
        if qkv is None:
            continue
        packed = getattr(qkv[0], "_devit_packed", (qkv[0],))[0]
        if getattr(packed, "_devit_arena", False):
            continue                      # tagged by the composite path: lives in the main stream's pool already
        if packed.untyped_storage().nbytes() <= packed.numel() * packed.element_size():    # its own allocation
            packed.record_stream(main)


def _side_forward(teacher_model, samples, main, side):
    side.wait_stream(main)
    prev, ops.ARENA_ALLOC_STREAM = ops.ARENA_ALLOC_STREAM, main
    try:
        with torch.cuda.stream(side), torch.no_grad(), de_vit.lean_tail(teacher_model):
            return teacher_model(samples, output_qkv=True)
    finally:
        ops.ARENA_ALLOC_STREAM = prev


def distill_forward(model, teacher_model, samples, targets, gama=(0.2, 0.1, 0.3), kind="hard", alpha=0.5, tau=1.0,
                    criterion=None, dp_scales="draw", teacher_outputs=None, after_student=None):
    """One DEKD forward.  Returns dict(loss, cls_loss, q_loss, k_loss, v_loss, logits, teacher_logits).
    teacher_outputs: the teacher's outputs for `samples` if they were computed ahead (TeacherLookahead).
    after_student: called once the student's forward is enqueued (TeacherLookahead.launch of the NEXT batch)."""
    vit = model.module if hasattr(model, "module") else model
    # only the middle block's q/k/v enter the relation loss (engine.py:91-100): the other blocks' packed qkv buffers
    # need no zeroed overhang rows
    for m in (vit, teacher_model):
        if getattr(m, "qkv_pad_layers", None) is None and hasattr(m, "blocks"):
            m.qkv_pad_layers = {len(m.blocks) // 2 - 1}
    pre_teacher = None
    if teacher_outputs is None and os.environ.get("DEVIT_TEACHER_STREAM", "1") == "1" and samples.is_cuda:
        pre_teacher = _teacher_forward_async(teacher_model, samples)
    # only the logits and the middle block's q/k/v are read below: the last block runs on its two token rows (de_vit.lean_tail)
    with de_vit.lean_tail(vit):
        if dp_scales != "draw":   # explicit DropPath masks (parity tests)
            outputs = _forward_with_dp(vit, samples, dp_scales)
        else:
            outputs = model(samples, output_qkv=True)                               # engine.py:70
    logits, qkvs = outputs['output'], outputs['qkv']
    if after_student is not None:
        after_student()
    if teacher_outputs is None:                                                     # engine.py:73-76
        teacher_outputs = pre_teacher() if pre_teacher is not None else _teacher_forward(teacher_model, samples)
    teacher_logits, teacher_qkvs = teacher_outputs['output'], teacher_outputs['qkv']
    if criterion is None:
        criterion = losses.DistillLoss(losses.SoftTargetCrossEntropy(), kind, alpha, tau)
    cls_loss = criterion(outputs=logits, teacher_outputs=teacher_logits, labels=targets)   # :79
    tl, sl = len(teacher_qkvs), len(qkvs)
    assert tl % sl == 0, 'The number of student layer can not be divisible by the number of teacher layer'
    rel3 = losses.relation_losses_vector(qkvs[sl // 2 - 1], teacher_qkvs[tl // 2 - 1])     # :91-100, as one [q, k, v] tensor
    if rel3 is not None:
        # :102-106 on the vector: total = cls + sum_j (gama_j / sl) loss_j (one multiply-sum + one add; backward one multiply) and the
        # three logged values as views of loss / sl -- the scalar form costs ~25 tiny kernels per step around the same numbers
        scaled = rel3 / sl
        q_loss, k_loss, v_loss = scaled[0], scaled[1], scaled[2]
        # (an elementwise product + sum, not torch.dot: on ROCm torch.dot is a rocBLAS call, and no vendor BLAS runs on this path)
        loss = cls_loss + (rel3 * _relation_weights(gama, sl, rel3.device)).sum()
    else:
        q_loss, k_loss, v_loss = losses.relation_losses_packed(qkvs[sl // 2 - 1], teacher_qkvs[tl // 2 - 1])
        q_loss, k_loss, v_loss = q_loss / sl, k_loss / sl, v_loss / sl                  # :102-104
        loss = cls_loss + float(gama[0]) * q_loss + float(gama[1]) * k_loss + float(gama[2]) * v_loss   # :105-106
    return dict(loss=loss, cls_loss=cls_loss, q_loss=q_loss, k_loss=k_loss, v_loss=v_loss, logits=logits,
                teacher_logits=teacher_logits)


_side_stream = {}
_rel_w = {}


def _relation_weights(gama, layers, device):
    """[gama_q, gama_k, gama_v] / layers as a cached device tensor."""
    key = (float(gama[0]), float(gama[1]), float(gama[2]), int(layers), str(device))
    w = _rel_w.get(key)
    if w is None:
        w = _rel_w[key] = torch.tensor([key[0] / layers, key[1] / layers, key[2] / layers], dtype=torch.float32, device=device)
    return w


def _teacher_forward(teacher_model, samples):
    """Frozen teacher forward.  With DEVIT_TEACHER_STREAM=1 it is enqueued on a side stream before the student
    forward has drained, so the tail rounds of one model's GEMMs are filled by the other's workgroups."""
    if os.environ.get("DEVIT_TEACHER_STREAM", "1") != "1" or not samples.is_cuda:
        with torch.no_grad(), de_vit.lean_tail(teacher_model):
            return teacher_model(samples, output_qkv=True)
    main = torch.cuda.current_stream()
    side = _side_stream.setdefault(samples.device.index, torch.cuda.Stream())
    out = _side_forward(teacher_model, samples, main, side)   # NOTE: call BEFORE the student forward is enqueued to overlap
    main.wait_stream(side)
    _hand_over(out, main)
    return out


def _teacher_forward_async(teacher_model, samples):
    """Enqueue the teacher forward on the side stream now; the returned callable joins it."""
    main = torch.cuda.current_stream()
    side = _side_stream.setdefault(samples.device.index, torch.cuda.Stream())
    out = _side_forward(teacher_model, samples, main, side)

    def join():
        main.wait_stream(side)
        _hand_over(out, main)
        return out
    return join


def row_dtypes_for(*models):
    """The 16-bit patch-row types one im2row pass must produce so that every model of the step can read it (ops.patch_rows(dtypes=...)); None when
    a model runs the exact-fp32 kernels (they read the images themselves) or is not on a GPU."""
    out = []
    for m in models:
        if m is None:
            continue
        prec = getattr(m, "precision", "bf16")
        if prec == "f32" or not any(p.is_cuda for p in m.parameters()):
            return None
        dt = torch.float16 if prec == "f16" else torch.bfloat16
        if dt not in out:
            out.append(dt)
    return tuple(out) or None


class TeacherLookahead:
    """Frozen-teacher forward one batch ahead (engine.py:73-76 computes it inside the step).

    The teacher has no dependence on the student, so its forward for batch k+1 can be enqueued on the side stream as
    soon as batch k+1 exists: it then runs beside the student's backward of batch k, and the workgroups of one model
    fill the partial last rounds of the other's kernels for the whole step, not only during the student forward.
    Same arithmetic per step; one batch of teacher outputs (logits + one block's q/k/v) stays resident.

        look = TeacherLookahead(teacher); look.submit(batch0)
        for k: t_out = look.take(batch_k); look.submit(batch_k+1); distill_forward(..., batch_k, teacher_outputs=t_out)
    """

    def __init__(self, teacher_model):
        self.teacher = teacher_model
        if getattr(teacher_model, "qkv_pad_layers", None) is None and hasattr(teacher_model, "blocks"):
            teacher_model.qkv_pad_layers = {len(teacher_model.blocks) // 2 - 1}
        self._pending = None

    def submit(self, samples, defer=False):
        """defer: record the batch, enqueue its forward at launch() (or at take(), at the latest).  Round 6: the loop launches the NEXT batch's
        teacher forward behind the student's forward of the current one (distill_forward(after_student=...)) instead of in front of it: the side
        stream then starts when the student forward has drained, and the teacher runs beside the backward -- whose launches leave CUs idle (198-tile
        dgrads, attention, LayerNorm) -- and the next forward's first half, not beside a forward of CU-filling GEMMs that it only delays.  Same
        arithmetic, +0.9 % (three interleaved pairs: 11372 / 11407 / 11395 -> 11487 / 11488 / 11552 img/s).  DEVIT_TEACHER_SUBMIT=early: as before."""
        if self._pending is not None:
            raise RuntimeError("TeacherLookahead.submit: the previous batch was not taken")
        self._pending = [samples, None]
        if not defer or os.environ.get("DEVIT_TEACHER_SUBMIT", "late") == "early":
            self.launch()

    def launch(self):
        if self._pending is not None and self._pending[1] is None:
            self._pending[1] = _teacher_forward_async(self.teacher, self._pending[0])

    def take(self, samples):
        if self._pending is None or self._pending[0] is not samples:
            raise RuntimeError("TeacherLookahead.take: not the batch that was submitted")
        self.launch()
        join, self._pending = self._pending[1], None
        return join()


class _PreparedBatches:
    """data_loader -> (samples on device after mixup, targets, teacher outputs or None), one batch ahead when a
    TeacherLookahead is given (engine.py:62-68 does the transfer + mixup at the top of each iteration)."""

    def __init__(self, loader, device, mixup_fn, look, row_dtypes=None):
        self.loader, self.device, self.mixup_fn, self.look = loader, device, mixup_fn, look
        # 16-bit types of the patch rows the step's models read (row_dtypes_for): a batch that arrives as fp32 images (no device mixup) is cut
        # into patch rows ONCE here, and the student and the look-ahead teacher share them, as they share the mixed rows when mixup is on
        self.row_dtypes = row_dtypes
        self._h2d = None

    def __len__(self):
        return len(self.loader)

    def _to_device(self, samples, targets):
        """Host batches cross PCIe on a copy stream of their own (154 MB per bs-256 batch, ~3 ms): _prep runs one
        batch ahead of the step that consumes it, and the host enqueues ahead of the GPU, so the copy lands under the
        previous step's kernels instead of in front of this one's (pinned source memory assumed for that)."""
        dev = torch.device(self.device)
        if dev.type != "cuda" or (samples.device.type != "cpu" and targets.device.type != "cpu"):
            return samples.to(dev, non_blocking=True), targets.to(dev, non_blocking=True)
        if self._h2d is None:
            self._h2d = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        with torch.cuda.stream(self._h2d):
            samples = samples.to(dev, non_blocking=True)
            targets = targets.to(dev, non_blocking=True)
        main.wait_stream(self._h2d)
        for t in (samples, targets):
            t.record_stream(main)
        return samples, targets

    def _prep(self, batch):
        samples, targets = batch
        samples, targets = self._to_device(samples, targets)
        if self.mixup_fn is not None:
            samples, targets = self.mixup_fn(samples, targets)
        if self.row_dtypes and isinstance(samples, torch.Tensor) and samples.is_cuda and samples.dim() == 4:
            samples = ops.patch_rows(samples, dtypes=self.row_dtypes)
        return samples, targets

    def launch_teacher(self):
        """distill_forward(after_student=...): the student's forward of the current batch is enqueued -- now the next batch's teacher forward"""
        if self.look is not None:
            self.look.launch()

    def __iter__(self):
        it = iter(self.loader)
        cur = next(it, None)
        if cur is None:
            return
        cur = self._prep(cur)
        if self.look is not None:
            self.look.submit(cur[0])
        while cur is not None:
            nxt = next(it, None)
            nxt = self._prep(nxt) if nxt is not None else None
            t_out = None
            if self.look is not None:
                t_out = self.look.take(cur[0])
                if nxt is not None:
                    self.look.submit(nxt[0], defer=True)      # launched by launch_teacher() behind the student's forward, or by the next take()
            yield cur[0], cur[1], t_out
            cur = nxt


def _forward_with_dp(vit, samples, dp_scales):
    x = vit.embed(samples)
    xo, qkvs, _, _ = de_vit.run_blocks(list(vit.blocks), x, vit.training, True, False, False,
                                       grad_ready=vit.grad_ready, dp_scales=dp_scales, precision=vit.precision,
                                       qkv_pad_layers=getattr(vit, "qkv_pad_layers", None),
                                       lean_tokens=vit.num_tokens if getattr(vit, "_lean_tail", False) else 0)
    heads = vit._tokens_and_logits(xo, True)
    return {'output': (heads[1], heads[2]) if vit.training else (heads[1] + heads[2]) / 2, 'qkv': qkvs}


def train_1epoch_qkv(model, teacher_model, criterion, data_loader, optimizer, device, epoch, loss_scaler, log, args,
                     max_norm=0, model_ema=None, mixup_fn=None, print_freq=10):
    """engine.py:48-140.  `loss_scaler(loss, optimizer, clip_grad=, parameters=)` keeps timm NativeScaler's call
    shape; bf16 needs no loss scaling, so devit_amd.optim.StepRunner is the usual object here."""
    from .utils import MetricLogger, SmoothedValue
    model.train(True)
    metric_logger = MetricLogger(delimiter="  ")
    for name in ('lr', 'cls_loss', 'q_loss', 'k_loss', 'v_loss'):
        metric_logger.add_meter(name, SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    step = 0
    lookahead = bool(getattr(args, "teacher_lookahead", True)) and device is not None and str(device).startswith("cuda")
    batches = _PreparedBatches(data_loader, device, mixup_fn, TeacherLookahead(teacher_model) if lookahead else None,
                               row_dtypes=row_dtypes_for(model, teacher_model))
    for samples, targets, teacher_out in metric_logger.log_every(batches, print_freq, header):
        out = distill_forward(model, teacher_model, samples, targets, gama=args.gama, criterion=criterion,
                              teacher_outputs=teacher_out, after_student=batches.launch_teacher)
        loss = out['loss']
        log_now = (step % print_freq == 0)
        if log_now:
            loss_value = loss.item()
            if not math.isfinite(loss_value):                                        # engine.py:119-121
                print("Loss is {}, stopping training".format(loss_value))
                sys.exit(1)
            metric_logger.update(cls_loss=out['cls_loss'].item(), q_loss=out['q_loss'].item(),
                                 k_loss=out['k_loss'].item(), v_loss=out['v_loss'].item(), loss=loss_value)
        optimizer.zero_grad()
        loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters())
        if model_ema is not None:
            model_ema.update(model)
        if log_now:
            metric_logger.update(lr=optimizer.param_groups[0]["lr"])
        step += 1
    metric_logger.synchronize_between_processes()
    if log is not None:
        log.info(f"Averaged stats: {metric_logger}")
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


def train_one_epoch(model, criterion, data_loader, optimizer, device, epoch, loss_scaler, log=None, max_norm=0,
                    model_ema=None, mixup_fn=None, print_freq=10):
    """train_subdata.py:233-286: the plain (teacher-in-criterion) training loop that produces the sub-dataset teachers
    and fine-tuned models -- `criterion` is losses.DistillationLoss(inputs, outputs, labels).  Same kernels as the DEKD
    step; `.item()` only every print_freq steps (the reference syncs every step, :266,:279)."""
    from .utils import MetricLogger, SmoothedValue
    model.train(True)
    metric_logger = MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    step = 0
    for samples, targets, _ in metric_logger.log_every(_PreparedBatches(data_loader, device, mixup_fn, None), print_freq, header):
        with de_vit.lean_tail(model):                                                # only the logits are read below
            outputs = model(samples)                                                 # :258
        loss = criterion(inputs=samples, outputs=outputs, labels=targets)            # :259
        if step % print_freq == 0:
            loss_value = loss.item()
            if not math.isfinite(loss_value):                                        # :263-265
                print("Loss is {}, stopping training".format(loss_value))
                sys.exit(1)
            metric_logger.update(loss=loss_value, lr=optimizer.param_groups[0]["lr"])
        optimizer.zero_grad()
        loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters())
        if model_ema is not None:
            model_ema.update(model)
        step += 1
    metric_logger.synchronize_between_processes()
    if log is not None:
        log.info(f"Averaged stats: {metric_logger}")
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


@torch.no_grad()
def evaluate(data_loader, model, device):
    """engine.py:17-45: eval forward, CE, top-1 / top-5."""
    from .utils import MetricLogger, accuracy
    metric_logger = MetricLogger(delimiter="  ")
    model.eval()
    for images, target in metric_logger.log_every(data_loader, 10, 'Test:'):
        images = images.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        with de_vit.lean_tail(model):                                   # only the logits are read (engine.py:31-32)
            output = model(images)
        loss = losses.DistillLoss(losses.SoftTargetCrossEntropy(), 'none', 0., 1.)(output, None, target)
        acc1, acc5 = accuracy(output, target, topk=(1, min(5, output.shape[1])))
        metric_logger.update(loss=loss.item())
        metric_logger.meters['acc1'].update(acc1.item(), n=images.shape[0])
        metric_logger.meters['acc5'].update(acc5.item(), n=images.shape[0])
    metric_logger.synchronize_between_processes()
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


# ------------------------------------------------------------------------------------------------------
# ensemble stage (engine.py:143-242): MultiViT backbones + EnsMLP fusion against the teacher
# ------------------------------------------------------------------------------------------------------
def ens_forward(model, ens_model, criterion, samples, targets, distillation_type="hard"):
    """engine.py:167-179: returns dict(loss, token_loss, cls_loss)."""
    features = model(samples)
    if distillation_type == 'none':
        logits = ens_model(features)
        loss = criterion(samples, logits, targets)
        return dict(loss=loss, token_loss=None, cls_loss=loss)
    outputs = ens_model(features, True)
    inter_loss, cls_loss = criterion(inputs=samples, stu_outputs=outputs, labels=targets)
    return dict(loss=inter_loss + cls_loss, token_loss=inter_loss, cls_loss=cls_loss)


def train_1epoch_ens_disjoint(model, ens_model, criterion, data_loader, optimizer, ens_optimizer, device, epoch, scaler,
                              args, log, model_ema=None, ens_model_ema=None, mixup_fn=None, max_norm=0, print_freq=10):
    """engine.py:143-210.  `optimizer` / `ens_optimizer` are any torch optimizers over the two parameter sets (the
    reference uses two AdamW instances); gradients arrive in param.grad."""
    from .utils import MetricLogger, SmoothedValue
    model.train(True)
    ens_model.train(True)
    metric_logger = MetricLogger(delimiter="  ")
    for name in ('backbone_lr', 'ens_lr', 'cls_loss', 'token_loss'):
        metric_logger.add_meter(name, SmoothedValue(window_size=1, fmt='{value:.6f}'))
    step = 0
    for samples, targets in metric_logger.log_every(data_loader, print_freq, 'Epoch: [{}]'.format(epoch)):
        samples, targets = samples.to(device, non_blocking=True), targets.to(device, non_blocking=True)
        if mixup_fn is not None:
            samples, targets = mixup_fn(samples, targets)
        optimizer.zero_grad(set_to_none=True)
        ens_optimizer.zero_grad(set_to_none=True)
        out = ens_forward(model, ens_model, criterion, samples, targets, args.distillation_type)
        out['loss'].backward()
        # the gradient mean DistributedDataParallel produces for both wrapped models (ensemble.py:332-334); no-op at world 1
        from . import ddp
        ddp.allreduce_mean_([p.grad for p in model.parameters()] + [p.grad for p in ens_model.parameters()])
        if max_norm:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
            torch.nn.utils.clip_grad_norm_(ens_model.parameters(), max_norm)
        optimizer.step()
        ens_optimizer.step()
        if step % print_freq == 0:
            lv = out['loss'].item()
            if not math.isfinite(lv):
                print(f"Loss is {lv}, stopping training")
                sys.exit(1)
            metric_logger.update(loss=lv, backbone_lr=optimizer.param_groups[0]["lr"], ens_lr=ens_optimizer.param_groups[0]["lr"])
            if out['token_loss'] is not None:
                metric_logger.update(token_loss=out['token_loss'].item(), cls_loss=out['cls_loss'].item())
        step += 1
    metric_logger.synchronize_between_processes()
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


@torch.no_grad()
def evaluate_ens_disjoint(data_loader, model, ens_model, device):
    """engine.py:212-242: collaborative inference = N sub-model backbones + EnsMLP, CE, top-1 / top-5."""
    from .utils import MetricLogger, accuracy
    metric_logger = MetricLogger(delimiter="  ")
    model.eval()
    ens_model.eval()
    for images, target in metric_logger.log_every(data_loader, 10, 'Test:'):
        images, target = images.to(device, non_blocking=True), target.to(device, non_blocking=True)
        output = ens_model(model(images))
        loss = losses.DistillLoss(losses.SoftTargetCrossEntropy(), 'none', 0., 1.)(output, None, target)
        acc1, acc5 = accuracy(output, target, topk=(1, 5))
        metric_logger.update(loss=loss.item())
        metric_logger.meters['acc1'].update(acc1.item(), n=images.shape[0])
        metric_logger.meters['acc5'].update(acc5.item(), n=images.shape[0])
    metric_logger.synchronize_between_processes()
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
