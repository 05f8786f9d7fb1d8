#!/usr/bin/env python3
"""train_subdata.py on MI355X: flag-compatible re-host of the reference's sub-dataset training CLI
(train_subdata.py:36-190 flags, :193-230 model loading, :320-470 main) -- the loop that fine-tunes a model (by default
the DeiT-B teacher) on one division of the dataset, optionally distilling from another teacher through DeiT's
DistillationLoss.  Same models, kernels, fused optimizer and `--synthetic N` data as distill_sub.py; defaults that
differ from distill_sub.py follow the reference (--model deit_base_distilled_patch16_224, --distillation-type none,
--weight-decay 0, --epochs 5).
"""
import argparse
import datetime
import json
import os
import time
from pathlib import Path

import numpy as np
import torch

import devit_amd
import distill_sub as ds
from devit_amd import ddp, engine, losses, optim, utils


def get_args_parser():
    p = argparse.ArgumentParser('DeViT sub-dataset training (MI355X)', add_help=False, parents=[ds.get_args_parser()],
                                conflict_handler='resolve')
    p.add_argument('--model', default='deit_base_distilled_patch16_224', type=str, metavar='MODEL')
    p.add_argument('--distillation-type', default='none', choices=['none', 'soft', 'hard'], type=str)
    p.add_argument('--epochs', default=5, type=int)
    p.add_argument('--weight-decay', type=float, default=0)
    p.add_argument('--no-aug', action='store_true', help='not use aug')            # train_subdata.py:112
    return p


def main(args):
    utils.init_distributed_mode(args)
    ds.check_supported(args)
    device = torch.device(args.device)
    seed = args.seed + utils.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    num_classes = ds.NUM_CLASSES[args.dataset] // args.num_division
    args.num_classes = num_classes
    if args.distillation_token:
        raise SystemExit("--distillation-token (resize_dim models) is outside the DeViT path")
    train_loader, val_loader, num_classes = ds.build_loaders(args, num_classes, device, provider="division")   # train_subdata.py:335-380
    args.num_classes = num_classes
    mixup_fn = None
    if args.mixup > 0 or args.cutmix > 0. or args.cutmix_minmax is not None:
        mixup_fn = ds.Mixup(args.mixup, args.cutmix, args.mixup_prob, args.mixup_switch_prob, args.smoothing, num_classes)

    # train_subdata.py:193-230
    model = devit_amd.create_model(args.model, pretrained=True, pretrained_path=args.model_path or None,
                                   num_classes=1000 if args.model_path else num_classes, drop_rate=args.drop,
                                   drop_path_rate=args.drop_path, drop_block_rate=None)
    if args.model_path:
        model.reset_classifier(num_classes=num_classes)
    model.to(device)
    teacher = None
    if args.distillation_type != 'none':
        teacher = devit_amd.create_model(args.teacher_model, num_classes=num_classes, drop_rate=args.drop,
                                         drop_path_rate=args.drop_path, drop_block_rate=None)
        tp = os.path.join(args.teacher_path, f'sub-dataset{args.start_division}', 'checkpoint.pth') if args.teacher_path else ''
        if tp and os.path.exists(tp):
            ck = torch.load(tp, map_location='cpu', weights_only=False)
            teacher.load_state_dict(ck['model'] if args.dataset == 'IMNET' and 'model' in ck else ck)
        teacher.to(device).eval()
        for p_ in teacher.parameters():
            p_.requires_grad_(False)
        teacher.request_precision(args.teacher_precision)
        if mixup_fn is not None:
            mixup_fn.set_precisions(model.precision, teacher.precision)

    flat = ddp.FlatParams(model)
    ddp.broadcast_parameters(flat)          # ranks are seeded seed + rank: rank 0's weights first, then the bf16 copies
    flat.attach_bf16(model)
    reducer = ddp.BucketedGradReducer(flat).attach(model)
    args.lr = args.lr * args.batch_size * utils.get_world_size() / 512.0             # train_subdata.py:404-405
    optimizer = optim.FlatAdamW(flat, lr=args.lr, eps=args.opt_eps, betas=tuple(args.opt_betas or (0.9, 0.999)),
                                weight_decay=args.weight_decay, max_norm=args.clip_grad,
                                ema_decay=args.model_ema_decay if args.model_ema else None,
                                no_decay=optim.no_decay_names(model))
    loss_scaler, lr_scheduler = ds.StepRunner(reducer), ds.CosineEpochs(optimizer, args)
    if mixup_fn is not None:                                                         # :409-416
        base = losses.SoftTargetCrossEntropy()
    elif args.smoothing:
        base = losses.LabelSmoothingCrossEntropy(smoothing=args.smoothing)
    else:
        base = torch.nn.CrossEntropyLoss()
    criterion = losses.DistillationLoss(base, teacher, args.distillation_type, args.distillation_alpha,
                                        args.distillation_tau, args.distillation_token)
    if args.eval:
        print(engine.evaluate(val_loader, model, device))
        return
    output_dir = Path(os.path.join(args.output_dir, f'sub-dataset{args.start_division}'))
    output_dir.mkdir(parents=True, exist_ok=True)
    max_accuracy, start = 0.0, time.time()
    for epoch in range(args.start_epoch, args.epochs):
        if args.distributed:
            ds.set_epoch(train_loader, epoch)
        train_stats = engine.train_one_epoch(model=model, criterion=criterion, data_loader=train_loader, optimizer=optimizer,
                                             device=device, epoch=epoch, loss_scaler=loss_scaler, max_norm=args.clip_grad,
                                             mixup_fn=mixup_fn)
        lr_scheduler.step(epoch)
        utils.save_on_master({'model': model.state_dict(), 'optimizer': optimizer.state_dict(),
                              'lr_scheduler': lr_scheduler.state_dict(), 'epoch': epoch,
                              'scaler': loss_scaler.state_dict(), 'args': args}, output_dir / 'checkpoint_temp.pth')
        test_stats = engine.evaluate(val_loader, model, device)
        print(f"Epoch: {epoch}/{args.epochs}  [Train] Loss: {train_stats.get('loss', float('nan')):.4f}  "
              f"[Eval] Top-1: {test_stats['acc1']:.4f} Top-5: {test_stats['acc5']:.4f} Loss: {test_stats['loss']:.4f}")
        if max_accuracy < test_stats["acc1"]:
            max_accuracy = test_stats["acc1"]
            if utils.is_main_process():
                torch.save(model.state_dict(), output_dir / 'checkpoint.pth')
        if utils.is_main_process():
            with (output_dir / "log.txt").open("a") as f:
                f.write(json.dumps({**{f'train_{k}': v for k, v in train_stats.items()},
                                    **{f'test_{k}': v for k, v in test_stats.items()}, 'epoch': epoch}) + "\n")
    print(f'Training time {datetime.timedelta(seconds=int(time.time() - start))} on sub-dataset{args.start_division}')


if __name__ == '__main__':
    parser = argparse.ArgumentParser('DeViT sub-dataset training (MI355X)', parents=[get_args_parser()])
    main(parser.parse_args())
