#!/usr/bin/env python3
"""ensemble.py on MI355X: flag-compatible re-host of the reference's collaborative-inference CLI (ensemble.py:37-189
flags, :192-242 model loading, :447-454 output-directory naming).  N distilled sub-models (MultiViT) + the EnsMLP
fusion head are trained against the teacher with EnsLoss and evaluated as one classifier (BASELINE config 5).

Flags shared with distill_sub.py keep their reference defaults except the ones ensemble.py changes (--lr 1e-5);
`--synthetic N` as in distill_sub.py.  Sub-model checkpoints are read from
`{--model-path}/sub-dataset{i}/checkpoint.pth` and copied positionally (all tensors but the heads), the teacher from
`--teacher-path`; with --synthetic and no paths the models are randomly initialised.
"""
import argparse
import json
import os
import time
from pathlib import Path

import numpy as np
import torch

import devit_amd
import distill_sub as ds
from devit_amd import ddp, engine, losses, optim, utils
from devit_amd.de_vit import model_config
from devit_amd.ensemble_models import EnsMLP, MultiViT, load_sub_checkpoints


def get_args_parser():
    base = ds.get_args_parser()
    p = argparse.ArgumentParser('DeViT Ensemble script (MI355X)', add_help=False, parents=[base], conflict_handler='resolve')
    # defaults where ensemble.py differs from distill_sub.py (ensemble.py:39-41,47,68,72,77)
    p.add_argument('--lr', type=float, default=1e-5, metavar='LR')
    p.add_argument('--epochs', default=3, type=int)
    p.add_argument('--eval-batch-size', default=10, type=int)
    p.add_argument('--clip-grad', type=float, default=None, metavar='NORM')
    p.add_argument('--weight-decay', type=float, default=0.05)
    p.add_argument('--model-path', type=str, default=r'./ckpt')
    p.add_argument('--no-aug', action='store_true', help='not use aug')
    p.add_argument('--loss', default='mse', choices=['mse', 'kldiv'], type=str, help="loss type")
    p.add_argument('--dataset', default='cifar100', choices=['cifar100', 'IMNET', 'INAT', 'INAT19'])
    p.add_argument('--sub_classes', nargs='+', default=[25, 25, 25, 25])
    p.add_argument('--gates', default='', help='file written by devit_amd.shrink.save_gates for the MultiViT '
                                               '(the reference drops the gates of shrunk sub-models on reload)')
    p.add_argument('--physical-shrink', action='store_true',
                   help='--eval only: remove the gated-off heads / neurons from the GEMMs (devit_amd.shrink.compact)')
    return p


def get_models(args, num_subs, sub_classes, num_classes):
    """ensemble.py:203-242."""
    teacher = None
    if args.distillation_type != 'none':
        teacher = devit_amd.create_model(args.teacher_model, num_classes=num_classes, drop_rate=args.drop,
                                         drop_path_rate=args.drop_path, drop_block_rate=None)
        if args.teacher_path and os.path.exists(args.teacher_path):
            teacher.load_state_dict(torch.load(args.teacher_path, map_location='cpu'))
        elif not args.synthetic:
            raise SystemExit(f"teacher checkpoint not found: {args.teacher_path}")
        teacher.to(args.device).eval()
        for p_ in teacher.parameters():
            p_.requires_grad_(False)
        teacher.request_precision(args.teacher_precision)
    model = MultiViT(model=args.model, drop=args.drop, drop_path=args.drop_path, num_div=num_subs, num_classes_list=sub_classes)
    # sub_size from the constructed backbones (the reference reads a wrong 192 from its config table)
    ens_model = EnsMLP(model=args.model, num_class=num_classes, sub_size=model.backbones[0].embed_dim,
                       num_classes_list=sub_classes, teacher_size=model_config[args.teacher_model]['embed_dim'])
    paths = [os.path.join(args.model_path or '', f'sub-dataset{i}', 'checkpoint.pth') for i in range(num_subs)]
    if all(os.path.exists(p_) for p_ in paths):
        load_sub_checkpoints(model, [torch.load(p_, map_location='cpu') for p_ in paths])
    elif not args.synthetic:
        raise SystemExit(f"sub-model checkpoints not found under {args.model_path}")
    return teacher, model.to(args.device), ens_model.to(args.device)


def param_groups(module, weight_decay):
    """timm create_optimizer's grouping (ensemble.py:341-342): no decay for 1-D tensors, biases, no_weight_decay()."""
    skip = optim.no_decay_names(module)
    named = [(n, p_) for n, p_ in module.named_parameters() if p_.requires_grad]
    if not weight_decay:
        return [{'params': [p_ for _, p_ in named], 'weight_decay': 0.0}]
    return [{'params': [p_ for n, p_ in named if n not in skip], 'weight_decay': weight_decay},
            {'params': [p_ for n, p_ in named if n in skip], 'weight_decay': 0.0}]


def main(args):
    utils.init_distributed_mode(args)
    ds.check_supported(args)
    device = torch.device(args.device)
    torch.manual_seed(args.seed + utils.get_rank())
    np.random.seed(args.seed + utils.get_rank())
    sub_classes = [int(c) for c in args.sub_classes]
    num_classes = sum(sub_classes)
    args.num_classes = num_classes
    train_loader, val_loader, _ = ds.build_loaders(args, num_classes, device, provider="whole", plain_sampler_over="train")   # ensemble.py:261-300
    mixup_fn = ds.Mixup(args.mixup, args.cutmix, args.mixup_prob, args.mixup_switch_prob, args.smoothing, num_classes) \
        if (args.mixup > 0 or args.cutmix > 0.) else None
    teacher, model, ens_model = get_models(args, len(sub_classes), sub_classes, num_classes)
    if mixup_fn is not None:        # the fused Mixup + im2row emits the patch rows in every 16-bit type a model of the step reads
        mixup_fn.set_precisions("bf16", getattr(teacher, "precision", None))
    if args.gates:
        from devit_amd import shrink
        shrink.load_gates(model, args.gates)
    if args.eval:
        if args.physical_shrink:
            from devit_amd import shrink
            rep = shrink.compact(model)
            print(f"physically shrunk {len(rep)} blocks: heads run {sorted(set(r[1] for r in rep))}, "
                  f"hidden widths run {sorted(set(r[3] for r in rep))}")
        print(engine.evaluate_ens_disjoint(val_loader, model, ens_model, device))
        return
    # ranks are seeded seed + rank: every replica starts from rank 0's weights, as under DistributedDataParallel
    ddp.broadcast_module(model)
    ddp.broadcast_module(ens_model)
    args.lr = args.lr * args.batch_size * utils.get_world_size() / 512.0                      # ensemble.py:339-340
    betas = tuple(args.opt_betas or (0.9, 0.999))
    optimizer = torch.optim.AdamW(param_groups(model, args.weight_decay), lr=args.lr, eps=args.opt_eps, betas=betas)
    ens_optimizer = torch.optim.AdamW(param_groups(ens_model, args.weight_decay), lr=args.lr, eps=args.opt_eps, betas=betas)
    lr_scheduler, ens_lr_scheduler = ds.CosineEpochs(optimizer, args), ds.CosineEpochs(ens_optimizer, args)   # :345-346
    if mixup_fn is not None:                 # ensemble.py:350-357: smoothing is handled by the mixup label transform
        base = losses.SoftTargetCrossEntropy()
    elif args.smoothing:
        base = losses.LabelSmoothingCrossEntropy(smoothing=args.smoothing)
    else:
        base = torch.nn.CrossEntropyLoss()
    criterion = losses.EnsLoss(base, teacher, args.model, args.distillation_type, args.distillation_alpha,
                               args.distillation_tau, args.loss)
    output_dir, max_accuracy, start = Path(args.output_dir), 0.0, time.time()
    for epoch in range(args.start_epoch, args.epochs):
        if args.distributed:
            ds.set_epoch(train_loader, epoch)
        train_stats = engine.train_1epoch_ens_disjoint(model, ens_model, criterion, train_loader, optimizer, ens_optimizer,
                                                       device, epoch, None, args, None, mixup_fn=mixup_fn,
                                                       max_norm=args.clip_grad)
        lr_scheduler.step(epoch)                                                                # ensemble.py:384-385
        ens_lr_scheduler.step(epoch)
        utils.save_on_master({'model': model.state_dict(), 'ens_model': ens_model.state_dict(),
                              'optimizer': optimizer.state_dict(), 'ens_optimizer': ens_optimizer.state_dict(),
                              'lr_scheduler': lr_scheduler.state_dict(), 'ens_lr_scheduler': ens_lr_scheduler.state_dict(),
                              'epoch': epoch, 'scaler': {}, 'args': args}, output_dir / 'checkpoint_temp.pth')
        test_stats = engine.evaluate_ens_disjoint(val_loader, model, ens_model, device)
        print(f"Epoch: {epoch}/{args.epochs} [Train] Loss: {train_stats.get('loss', float('nan')):.4f} "
              f"[Eval] Top-1: {test_stats['acc1']:.4f} Top-5: {test_stats['acc5']:.4f}")
        if max_accuracy < test_stats["acc1"] and utils.is_main_process():
            max_accuracy = test_stats["acc1"]
            torch.save({'model': model.state_dict(), 'ens_model': ens_model.state_dict()}, output_dir / 'checkpoint.pth')
        if utils.is_main_process():
            with (output_dir / "log.txt").open("a") as f:
                f.write(json.dumps({**{f'train_{k}': v for k, v in train_stats.items()},
                                    **{f'test_{k}': v for k, v in test_stats.items()}, 'epoch': epoch}) + "\n")
    print(f'Training time {int(time.time() - start)} s')


if __name__ == '__main__':
    parser = argparse.ArgumentParser('DeViT Ensemble script (MI355X)', parents=[get_args_parser()], conflict_handler='resolve')
    args = parser.parse_args()
    args.name = f'lr{args.lr}-bs{args.batch_size}-epochs{args.epochs}-grad{args.clip_grad}-wd{args.weight_decay}-wm{args.warmup_epochs}'
    method = {'none': 'sub_no_distill', 'soft': 'distill_sub_soft', 'hard': 'distill_sub_hard'}
    args.method = f'ens_disjoint_{method[args.distillation_type]}_{args.loss}'
    args.output_dir = os.path.join(args.output_dir, f'{args.dataset}_div{args.num_division}', f'{args.model}', args.method, args.name)
    Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    main(args)
