#!/usr/bin/env python3
"""distill_sub.py on MI355X: flag-compatible re-host of the reference's DEKD distillation CLI.

Same flags, defaults, output-directory naming, checkpoint files and per-epoch loop as the reference
(distill_sub.py:36-205 flags, :247-251 naming, :412-469 loop); the step itself runs on devit_amd
(HIP kernels + bucketed RCCL gradient exchange + fused AdamW/EMA).  Launch like the reference:

    python -m torch.distributed.run --nproc_per_node=8 distill_sub.py --model dedeit \
        --teacher-model deit_base_distilled_patch16_224 --teacher-path <dir> --dataset cifar100 --num_division 4 ...

Differences, all host-side: (1) `--synthetic N` trains/evaluates on N on-device random batches per epoch (the image
has no torchvision / datasets; the JPEG pipeline of data/ is outside the hot path) -- without it the
reference's dataset package must be importable as `data.get_dataset` (its `build_division_dataset` / `build_dataset`);
samplers and loaders are then built as the reference builds them (`build_loaders`);
(2) bf16 needs no loss scaling: `scaler` in checkpoints is an empty dict; (3) Mixup/CutMix run as device-side
tensor ops (SURVEY §8f-4 "next").
"""
import argparse
import datetime
import json
import math
import os
import time
from pathlib import Path

import numpy as np
import torch

import devit_amd
from devit_amd import ddp, engine, losses, optim, utils
from devit_amd.de_vit import model_config


def get_args_parser():
    p = argparse.ArgumentParser('DeViT sub-model distillation (MI355X)', add_help=False)
    a = p.add_argument
    a('--batch-size', default=2, type=int); a('--eval-batch-size', default=512, type=int); a('--epochs', default=5, type=int)
    a('--output_dir', default=r'./output'); a('--finetune', action='store_true')
    a('--model', default='dedeit', type=str, metavar='MODEL'); a('--model-path', type=str, default='')
    a('--input-size', default=224, type=int); a('--drop', type=float, default=0.0); a('--drop-path', type=float, default=0.1)
    a('--model-ema', action='store_true'); a('--no-model-ema', action='store_false', dest='model_ema'); p.set_defaults(model_ema=True)
    a('--model-ema-decay', type=float, default=0.99996); a('--model-ema-force-cpu', action='store_true', default=False)
    a('--opt', default='adamw', type=str); a('--opt-eps', default=1e-8, type=float); a('--opt-betas', default=None, type=float, nargs='+')
    a('--clip-grad', type=float, default=1.0); a('--momentum', type=float, default=0.9); a('--weight-decay', type=float, default=0)
    a('--sched', default='cosine', type=str); a('--lr', type=float, default=5e-4)
    a('--lr-noise', type=float, nargs='+', default=None); a('--lr-noise-pct', type=float, default=0.67); a('--lr-noise-std', type=float, default=1.0)
    a('--warmup-lr', type=float, default=1e-6); a('--min-lr', type=float, default=1e-5); a('--decay-epochs', type=float, default=30)
    a('--warmup-epochs', type=int, default=5); a('--cooldown-epochs', type=int, default=10); a('--patience-epochs', type=int, default=10)
    a('--decay-rate', '--dr', type=float, default=0.1)
    a('--color-jitter', type=float, default=0.4); a('--aa', type=str, default='rand-m9-mstd0.5-inc1'); a('--smoothing', type=float, default=0.1)
    a('--train-interpolation', type=str, default='bicubic'); a('--repeated-aug', action='store_true')
    a('--no-repeated-aug', action='store_false', dest='repeated_aug'); p.set_defaults(repeated_aug=True); a('--no_aug', action='store_true')
    a('--reprob', type=float, default=0.25); a('--remode', type=str, default='pixel'); a('--recount', type=int, default=1)
    a('--resplit', action='store_true', default=False)
    a('--mixup', type=float, default=0.8); a('--cutmix', type=float, default=1.0); a('--cutmix-minmax', type=float, nargs='+', default=None)
    a('--mixup-prob', type=float, default=1.0); a('--mixup-switch-prob', type=float, default=0.5); a('--mixup-mode', type=str, default='batch')
    a('--teacher-model', default='vit_large_patch16_224', type=str); a('--teacher-path', type=str, default='')
    a('--distillation-type', default='hard', choices=['none', 'soft', 'hard'], type=str)
    a('--distillation-inter', type=bool, default=True); a('--distillation-token', action='store_true')
    a('--distillation-alpha', default=0.5, type=float); a('--distillation-tau', default=1.0, type=float)
    a('--gama', nargs='+', default=[0.2, 0.1, 0.3])
    a('--data-path', default=r'./datasets'); a('--dataset', default='cifar100', choices=['cifar100', 'IMNET', 'cars', 'pets', 'flowers'])
    a('--inat-category', default='name'); a('--num_division', metavar='N', type=int, default=4); a('--start-division', metavar='N', type=int, default=0)
    a('--device', default='cuda'); a('--seed', default=0, type=int); a('--resume', default=''); a('--start_epoch', default=0, type=int)
    a('--eval', action='store_true'); a('--dist-eval', action='store_true', default=False); a('--num_workers', default=4, type=int)
    a('--local_rank', type=int, default=-1); a('--pin-mem', action='store_true'); a('--no-pin-mem', action='store_false', dest='pin_mem')
    p.set_defaults(pin_mem=True); a('--world_size', default=1, type=int); a('--dist_url', default='env://')
    a('--load_shrink', action='store_true', default=False); a('--shrink_checkpoint', type=str, default='')
    a('--neuron_shrinking', action='store_true', default=False); a('--head_shrinking', action='store_true', default=False)
    a('--no-physical-shrink', dest='physical_shrink', action='store_false', default=True,
      help='train the gated student MASKED at the dense cost, as the reference does (default: through compacted blocks, '
           'devit_amd.shrink.compact(trainable=True): same function and gradients, the shrunk model\'s FLOPs)')
    a('--synthetic', type=int, default=0, metavar='STEPS', help='train on STEPS random on-device batches per epoch')
    a('--teacher-precision', default='bf16', choices=['f16', 'bf16'],
      help="16-bit type of the frozen teacher's forward (f16: teacher logits 1.1e-3 instead of 6.8e-3 from fp32; step 1.4 %% slower)")
    a('--no-teacher-lookahead', dest='teacher_lookahead', action='store_false',
      help='run the frozen teacher inside the step instead of one batch ahead (engine.TeacherLookahead)')
    return p


NUM_CLASSES = {'cifar100': 100, 'IMNET': 1000, 'cars': 196, 'pets': 37, 'flowers': 102}


class SyntheticLoader:
    """`steps` resident batches; same (images fp32 [B,3,224,224], labels int64 [B]) contract as the DataLoader."""

    def __init__(self, steps, batch, classes, device, seed):
        g = torch.Generator(device=device).manual_seed(seed)
        self.img = torch.randn((batch, 3, 224, 224), generator=g, device=device)
        self.lab = torch.randint(0, classes, (batch,), generator=g, device=device)
        self.steps = steps

    def __len__(self):
        return self.steps

    def __iter__(self):
        # fresh tensors every step, like a DataLoader: Mixup's cutmix branch writes into its input in place, and the
        # look-ahead teacher forward of step k+1 is still reading its batch while step k runs
        for _ in range(self.steps):
            yield self.img.clone(), self.lab.clone()


class RASampler(torch.utils.data.Sampler):
    """Repeated-augmentation sampler of utils/samplers.py:8-63 (DeiT's): every index `num_repeats` times in a row, the
    repeats of one sample landing on different ranks (rank r takes positions r, r + world, ...), an epoch truncated to
    floor(len // 256 * 256 / world) draws per rank; the permutation is seeded with the epoch."""

    def __init__(self, dataset, num_replicas, rank, shuffle=True, num_repeats=3):
        if num_repeats < 1:
            raise ValueError("num_repeats should be greater than 0")
        self.n, self.world, self.rank, self.shuffle, self.repeats, self.epoch = len(dataset), num_replicas, rank, shuffle, num_repeats, 0
        self.num_samples = int(math.ceil(self.n * num_repeats / num_replicas))
        self.total_size = self.num_samples * num_replicas
        self.num_selected_samples = int(math.floor(self.n // 256 * 256 / num_replicas))

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            idx = torch.randperm(self.n, generator=g)
        else:
            idx = torch.arange(self.n)
        idx = torch.repeat_interleave(idx, repeats=self.repeats, dim=0).tolist()
        idx += idx[:self.total_size - len(idx)]
        idx = idx[self.rank:self.total_size:self.world]
        return iter(idx[:self.num_selected_samples])

    def __len__(self):
        return self.num_selected_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


def build_loaders(args, num_classes, device, provider="division", plain_sampler_over="test"):
    """(train_loader, val_loader, num_classes).  `--synthetic N`: resident random batches.  Otherwise the reference's dataset
    package must be importable as `data.get_dataset` (its transforms need torchvision + timm, which this image lacks):
    `build_division_dataset(dataset_path=<data-path>/sub-dataset<k>, args=args)` for distill_sub / train_subdata
    (distill_sub.py:269-272), `build_dataset(args)` for ensemble (ensemble.py:261); samplers and loaders as
    distill_sub.py:274-313 builds them -- RASampler by default, the val loader sequential unless --dist-eval, train
    drop_last.  `plain_sampler_over`: with --no-repeated-aug distill_sub.py:281-283 and train_subdata.py:347-349 build the
    TRAIN sampler over the TEST set (its length sets the epoch length, and with it the LR schedule); kept, ensemble.py:271-273
    uses the train set."""
    if args.synthetic > 0:
        return (SyntheticLoader(args.synthetic, args.batch_size, num_classes, device, 1234 + utils.get_rank()),
                SyntheticLoader(max(1, args.synthetic // 8), args.batch_size, num_classes, device, 99), num_classes)
    try:
        import importlib
        gd = importlib.import_module("data.get_dataset")
    except Exception as e:
        raise SystemExit("no dataset provider importable (data.get_dataset, the reference's package: needs torchvision + timm); "
                         "use --synthetic N. " + repr(e))
    if provider == "division":
        train_ds, test_ds, num_classes = gd.build_division_dataset(
            dataset_path=os.path.join(args.data_path, f"sub-dataset{args.start_division}"), args=args)
    else:
        train_ds, test_ds, num_classes = gd.build_dataset(args)
    world, rank = utils.get_world_size(), utils.get_rank()
    if args.repeated_aug:
        sampler_train = RASampler(train_ds, num_replicas=world, rank=rank, shuffle=True)
    else:
        sampler_train = torch.utils.data.DistributedSampler(test_ds if plain_sampler_over == "test" else train_ds,
                                                            num_replicas=world, rank=rank, shuffle=True)
    if args.dist_eval:
        if len(test_ds) % world != 0:
            print("Warning: distributed evaluation with an eval set not divisible by the process count: duplicate entries are "
                  "added to equalise the ranks, which slightly alters the validation results.")
        sampler_val = torch.utils.data.DistributedSampler(test_ds, num_replicas=world, rank=rank, shuffle=False)
    else:
        sampler_val = torch.utils.data.SequentialSampler(test_ds)
    train_loader = torch.utils.data.DataLoader(train_ds, sampler=sampler_train, batch_size=args.batch_size,
                                               num_workers=args.num_workers, pin_memory=args.pin_mem, drop_last=True)
    val_loader = torch.utils.data.DataLoader(test_ds, sampler=sampler_val, batch_size=args.eval_batch_size,
                                             num_workers=args.num_workers, pin_memory=args.pin_mem, drop_last=False)
    return train_loader, val_loader, num_classes


def set_epoch(loader, epoch):
    """distill_sub.py:413-414: the distributed samplers reshuffle per epoch."""
    sampler = getattr(loader, "sampler", None)
    if hasattr(sampler, "set_epoch"):
        sampler.set_epoch(epoch)


class Mixup:
    """timm.data.Mixup(mode='batch') semantics (distill_sub.py:315-318, applied at engine.py:65-66).
    The draws (mixup vs cutmix, lambda ~ Beta, the box) are host-side numpy RNG as in timm; the arithmetic runs in two HIP
    kernels: devit_mix_im2row_bf16 turns the fp32 batch straight into the MIXED batch's bf16 patch rows (the mixed fp32
    images never exist; student and teacher both read those rows) and devit_mix_targets builds the soft targets."""

    def __init__(self, mixup_alpha, cutmix_alpha, prob, switch_prob, label_smoothing, num_classes, precisions=("bf16",)):
        self.ma, self.ca, self.prob, self.sw, self.eps, self.C = mixup_alpha, cutmix_alpha, prob, switch_prob, label_smoothing, num_classes
        self.set_precisions(*precisions)

    def set_precisions(self, *precisions):
        """The `precision` of every model that will read the mixed batch (student, teacher): which 16-bit patch rows the
        fused kernel emits ("bf16" / "f16"); with an "f32" model the mixed batch stays an fp32 image tensor."""
        self.precisions = tuple(p for p in precisions if p is not None) or ("bf16",)
        self.row_dtypes = tuple(dict.fromkeys(torch.float16 if p == "f16" else torch.bfloat16 for p in self.precisions if p != "f32"))

    def draw(self, H=224, W=224):
        """(mode, lam, box): mode 0 none / 1 mixup / 2 cutmix; lam already corrected to the clipped box area for cutmix."""
        lam, cut = 1.0, False
        if np.random.rand() < self.prob:
            cut = self.ca > 0 and (self.ma <= 0 or np.random.rand() < self.sw)
            lam = float(np.random.beta(self.ca, self.ca) if cut else np.random.beta(self.ma, self.ma))
        if cut:
            r = math.sqrt(1 - lam)
            ch, cw, cy, cx = int(H * r), int(W * r), np.random.randint(H), np.random.randint(W)
            y0, y1, x0, x1 = max(cy - ch // 2, 0), min(cy + ch // 2, H), max(cx - cw // 2, 0), min(cx + cw // 2, W)
            return 2, 1.0 - (y1 - y0) * (x1 - x0) / float(H * W), (y0, y1, x0, x1)
        return (1 if lam != 1.0 else 0), lam, (0, 0, 0, 0)

    def __call__(self, x, y):
        from devit_amd import ops
        assert x.shape[0] % 2 == 0, 'Batch size should be even when using this'
        mode, lam, box = self.draw(x.shape[-2], x.shape[-1])
        if "f32" in self.precisions or tuple(x.shape[-2:]) != (224, 224):
            # the exact-fp32 parity models read fp32 images (and the patch-row kernel is built for 224 x 224): timm's formulas on
            # the image tensor itself, mixed in place like timm does
            flipped = x.flip(0)
            if mode == 1:
                x.mul_(lam).add_(flipped, alpha=1.0 - lam)
            elif mode == 2:
                y0, y1, x0, x1 = box
                x[:, :, y0:y1, x0:x1] = flipped[:, :, y0:y1, x0:x1]
            return x, ops.mix_targets(y, self.C, lam, self.eps)
        return ops.mix_patch_rows(x, mode, lam, box, dtypes=self.row_dtypes), ops.mix_targets(y, self.C, lam, self.eps)


class CosineEpochs:
    """timm cosine scheduler as configured by create_scheduler: per-epoch, linear warm-up."""

    def __init__(self, opt, args):
        self.opt, self.a, self.base, self.last = opt, args, args.lr, -1
        self._set(0)

    def _set(self, epoch):
        a = self.a
        if epoch < a.warmup_epochs:
            lr = a.warmup_lr + (self.base - a.warmup_lr) * epoch / max(a.warmup_epochs, 1)
        else:
            lr = a.min_lr + 0.5 * (self.base - a.min_lr) * (1 + math.cos(math.pi * epoch / max(a.epochs, 1)))
        for g in self.opt.param_groups:
            g['lr'] = lr

    def step(self, epoch):
        """timm Scheduler.step(epoch) as called after each epoch (distill_sub.py:421): the LR becomes f(epoch), so the
        schedule runs one epoch behind the epoch counter -- epochs 0 and 1 both train at f(0) = warmup_lr.  Kept."""
        self.last = epoch
        self._set(epoch)

    def state_dict(self):
        return {'last': self.last}

    def load_state_dict(self, sd):
        self.last = sd['last']
        self._set(max(self.last, 0))


class StepRunner:
    """Call-compatible stand-in for timm NativeScaler (engine.py:127): backward, join the bucketed all-reduce,
    clip + AdamW + EMA in the fused optimizer.  bf16: nothing to scale."""

    def __init__(self, reducer):
        self.reducer = reducer

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False):
        loss.backward()
        self.reducer.finish()            # buckets hold sums; flat.grad_scale = 1 / world goes into the optimizer kernel
        optimizer.max_norm = clip_grad
        optimizer.step()

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass


def check_supported(args):
    """Flags the reference hands to timm factories that this build implements for one value only: refuse the others
    instead of silently training something else (create_optimizer / create_scheduler, distill_sub.py:340-343)."""
    if args.opt.lower() != 'adamw':
        raise SystemExit(f"--opt {args.opt}: only adamw (the reference's default) is built on the fused optimizer kernel")
    if args.sched != 'cosine':
        raise SystemExit(f"--sched {args.sched}: only cosine (the reference's default) is implemented")
    if args.lr_noise is not None:
        raise SystemExit("--lr-noise is not implemented")
    if getattr(args, 'distillation_token', False):
        raise SystemExit("--distillation-token (resize_dim models, models/de_vit.py:198-201) is outside the DEKD path")


def main(args):
    utils.init_distributed_mode(args)
    check_supported(args)
    args.method = 'distill_sub'
    args.name = (f'lr{args.lr}-bs{args.batch_size}-epochs{args.epochs}-grad{args.clip_grad}'
                 f'-wd{args.weight_decay}-wm{args.warmup_epochs}-gama{args.gama[0]}_{args.gama[1]}_{args.gama[2]}')
    args.output_dir = os.path.join(args.output_dir, f'{args.dataset}_div{args.num_division}', f'{args.model}', args.method, args.name)
    Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    device = torch.device(args.device)
    seed = args.seed + utils.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    num_classes = NUM_CLASSES[args.dataset] // args.num_division
    args.num_classes = num_classes
    train_loader, val_loader, num_classes = build_loaders(args, num_classes, device, provider="division")
    args.num_classes = num_classes

    mixup_fn = None
    if args.mixup > 0 or args.cutmix > 0. or args.cutmix_minmax is not None:
        mixup_fn = Mixup(args.mixup, args.cutmix, args.mixup_prob, args.mixup_switch_prob, args.smoothing, num_classes)

    stu_nb = 1000 if args.model_path != '' else num_classes
    resize_dim = model_config[args.teacher_model]["embed_dim"] if args.distillation_token else None
    model = devit_amd.create_model(args.model, pretrained=True, pretrained_path=args.model_path if args.finetune else None,
                                   num_classes=stu_nb, resize_dim=resize_dim, drop_rate=args.drop,
                                   drop_path_rate=args.drop_path, drop_block_rate=None)
    if args.model_path != '':
        model.reset_classifier(num_classes=num_classes)
    model.to(device)
    teacher = None
    if args.distillation_type != 'none':
        teacher = devit_amd.create_model(args.teacher_model, num_classes=num_classes, drop_rate=args.drop,
                                         drop_path_rate=args.drop_path, drop_block_rate=None)
        tp = os.path.join(args.teacher_path, f'sub-dataset{args.start_division}', 'checkpoint.pth') if args.teacher_path else ''
        if tp and os.path.exists(tp):
            teacher.load_state_dict(torch.load(tp, map_location='cpu'))
        elif not args.synthetic:
            raise SystemExit(f"teacher checkpoint not found: {tp}")
        teacher.to(device).eval()
        for p_ in teacher.parameters():
            p_.requires_grad_(False)
        teacher.request_precision(args.teacher_precision)
        if mixup_fn is not None:                    # Mixup's fused im2row emits the patch rows in every 16-bit type a model reads
            mixup_fn.set_precisions(model.precision, teacher.precision)

    flat = ddp.FlatParams(model)
    ddp.broadcast_parameters(flat)          # ranks are seeded seed + rank: rank 0's weights first, then the bf16 copies
    flat.attach_bf16(model)
    reducer = ddp.BucketedGradReducer(flat).attach(model)
    args.lr = args.lr * args.batch_size * utils.get_world_size() / 512.0             # distill_sub.py:338
    optimizer = optim.FlatAdamW(flat, lr=args.lr, eps=args.opt_eps, betas=tuple(args.opt_betas or (0.9, 0.999)),
                                weight_decay=args.weight_decay, max_norm=args.clip_grad,
                                ema_decay=args.model_ema_decay if args.model_ema else None,
                                no_decay=optim.no_decay_names(model))       # timm's two parameter groups
    loss_scaler, lr_scheduler = StepRunner(reducer), CosineEpochs(optimizer, args)
    if mixup_fn is not None:                 # distill_sub.py:345-352: smoothing is handled by the mixup label transform
        base = losses.SoftTargetCrossEntropy()
    elif args.smoothing:
        base = losses.LabelSmoothingCrossEntropy(smoothing=args.smoothing)
    else:
        base = torch.nn.CrossEntropyLoss()
    criterion = losses.DistillLoss(base, args.distillation_type, args.distillation_alpha, args.distillation_tau)
    n_parameters = sum(p_.numel() for p_ in model.parameters() if p_.requires_grad)

    if args.resume:
        ck = torch.load(args.resume, map_location='cpu', weights_only=False)   # holds the argparse Namespace, like the reference's
        model.load_state_dict(ck['model'])
        flat.refresh_bf16()
        if not args.eval and not args.finetune and 'optimizer' in ck and 'lr_scheduler' in ck and 'epoch' in ck:
            optimizer.load_state_dict(ck['optimizer'])
            lr_scheduler.load_state_dict(ck['lr_scheduler'])
            args.start_epoch = ck['epoch'] + 1
    if args.eval:
        print(engine.evaluate(val_loader, model, device))
        return

    # distill_sub.py:383-401: gate the student from a shrink policy before training
    if args.shrink_checkpoint or args.neuron_shrinking or args.head_shrinking:
        from devit_amd import shrink
        policy = shrink.apply_shrink(model, train_loader, args.shrink_checkpoint, args.neuron_shrinking,
                                     args.head_shrinking, device)
        if policy is not None:
            print("shrink: heads kept per block", [int(h.sum()) for h, _ in policy],
                  "neurons kept per block", [int(n.sum()) for _, n in policy])
            if args.physical_shrink:     # train at the shrunk model's FLOPs (the reference trains the masked model at the dense cost)
                rep = shrink.compact(model, trainable=True)
                print("shrink: compacted for training, (heads run, neurons run) per block", [(r[1], r[3]) for r in rep],
                      f"-> {shrink.compacted_gflops(model, num_classes=num_classes):.3f} GFLOP per image forward")

    # distill_sub.py:403-404: everything of this division goes under sub-dataset{start_division}/ -- where ensemble.py
    # (:228) looks for `{model-path}/sub-dataset{i}/checkpoint.pth`
    output_dir, max_accuracy, start = Path(args.output_dir) / f'sub-dataset{args.start_division}', 0.0, time.time()
    output_dir.mkdir(parents=True, exist_ok=True)
    for epoch in range(args.start_epoch, args.epochs):
        if args.distributed:
            set_epoch(train_loader, epoch)
        train_stats = engine.train_1epoch_qkv(model=model, teacher_model=teacher, criterion=criterion, args=args,
                                              data_loader=train_loader, optimizer=optimizer, device=device, epoch=epoch,
                                              loss_scaler=loss_scaler, log=None, max_norm=args.clip_grad, mixup_fn=mixup_fn)
        lr_scheduler.step(epoch)
        utils.save_on_master({'model': model.state_dict(), 'optimizer': optimizer.state_dict(),
                              'lr_scheduler': lr_scheduler.state_dict(), 'epoch': epoch,
                              'model_ema': optimizer.ema_state_dict(model), 'scaler': loss_scaler.state_dict(),
                              'args': args}, output_dir / 'checkpoint_temp.pth')
        test_stats = engine.evaluate(val_loader, model, device)
        print(f"Epoch: {epoch}/{args.epochs}  [Train] Loss: {train_stats.get('loss', float('nan')):.4f}  "
              f"[Eval] Top-1: {test_stats['acc1']:.4f} Top-5: {test_stats['acc5']:.4f} Loss: {test_stats['loss']:.4f}")
        if max_accuracy < test_stats["acc1"]:
            max_accuracy = test_stats["acc1"]
            if utils.is_main_process():
                torch.save(model.state_dict(), output_dir / 'checkpoint.pth')
                torch.save(args, output_dir / 'training_args.bin')
                (output_dir / 'result.txt').write_text(f'Final Accuracy: {max_accuracy}\n')
                if args.shrink_checkpoint:       # the gates are not in state_dict(): side-car file
                    from devit_amd import shrink
                    shrink.save_gates(model, output_dir / 'gates.pt')
        if utils.is_main_process():
            with (output_dir / "log.txt").open("a") as f:
                f.write(json.dumps({**{f'train_{k}': v for k, v in train_stats.items()},
                                    **{f'test_{k}': v for k, v in test_stats.items()}, 'epoch': epoch,
                                    'n_parameters': n_parameters}) + "\n")
    print(f'Training time {datetime.timedelta(seconds=int(time.time() - start))} on sub-dataset{args.start_division}')


if __name__ == '__main__':
    parser = argparse.ArgumentParser('DeViT sub-model distillation (MI355X)', parents=[get_args_parser()])
    main(parser.parse_args())
