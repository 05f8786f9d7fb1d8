"""The re-hosted distill_sub.py CLI runs end to end on one MI355X (synthetic data) and writes the reference's files."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_distill_sub_cli(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import argparse
    import distill_sub
    parser = argparse.ArgumentParser(parents=[distill_sub.get_args_parser()])
    args = parser.parse_args(["--synthetic", "4", "--batch-size", "4", "--epochs", "1", "--model", "dedeit",
                              "--teacher-model", "deit_base_distilled_patch16_224", "--dataset", "cifar100",
                              "--num_division", "4", "--output_dir", str(tmp_path), "--warmup-epochs", "0"])
    distill_sub.main(args)
    out = os.path.join(args.output_dir, "sub-dataset0")          # distill_sub.py:403-404
    assert os.path.exists(os.path.join(out, "checkpoint_temp.pth")) and os.path.exists(os.path.join(out, "log.txt"))
    ck = torch.load(os.path.join(out, "checkpoint_temp.pth"), map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "epoch", "model_ema", "scaler", "args"}
    assert len(ck["model"]) == 155 and len(ck["model_ema"]) == 155
    import json
    line = json.loads(open(os.path.join(out, "log.txt")).read().splitlines()[-1])
    assert line["train_loss"] == line["train_loss"] and line["n_parameters"] == 21685682   # finite, C = 25


@pytest.mark.parametrize("extra,expect", [([], "SoftTargetCrossEntropy"),
                                          (["--mixup", "0", "--cutmix", "0"], "LabelSmoothingCrossEntropy"),
                                          (["--mixup", "0", "--cutmix", "0", "--smoothing", "0"], "CrossEntropyLoss")])
def test_distill_sub_base_criterion_selection(tmp_path, monkeypatch, extra, expect):
    """distill_sub.py:345-352: SoftTargetCrossEntropy with mixup, LabelSmoothingCrossEntropy(--smoothing) without it,
    nn.CrossEntropyLoss with neither -- and the step runs with each."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import argparse
    import json
    import distill_sub
    from devit_amd import losses
    seen = []
    real = losses.DistillLoss

    class Rec(real):
        def __init__(self, base, *a, **k):
            seen.append(base)
            super().__init__(base, *a, **k)
    monkeypatch.setattr(losses, "DistillLoss", Rec)
    parser = argparse.ArgumentParser(parents=[distill_sub.get_args_parser()])
    args = parser.parse_args(["--synthetic", "2", "--batch-size", "4", "--epochs", "1", "--model", "dedeit",
                              "--teacher-model", "deit_base_distilled_patch16_224", "--dataset", "cifar100",
                              "--num_division", "4", "--output_dir", str(tmp_path), "--warmup-epochs", "0"] + extra)
    distill_sub.main(args)
    assert type(seen[0]).__name__ == expect
    if expect == "LabelSmoothingCrossEntropy":
        assert seen[0].smoothing == 0.1
    line = json.loads(open(os.path.join(args.output_dir, "sub-dataset0", "log.txt")).read().splitlines()[-1])
    assert line["train_loss"] == line["train_loss"]


def test_distill_sub_resume(tmp_path):
    """--resume (distill_sub.py:372-388): model, optimizer moments, EMA, step count and schedule come back from
    checkpoint_temp.pth and training continues at the next epoch."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import argparse
    import json
    import distill_sub
    base = ["--synthetic", "8", "--batch-size", "4", "--model", "dedeit", "--teacher-model",
            "deit_base_distilled_patch16_224", "--dataset", "cifar100", "--num_division", "4", "--warmup-epochs", "0",
            "--model-ema"]
    parse = lambda extra: argparse.ArgumentParser(parents=[distill_sub.get_args_parser()]).parse_args(base + extra)
    a1 = parse(["--epochs", "1", "--output_dir", str(tmp_path / "a")])
    distill_sub.main(a1)
    ck_path = os.path.join(a1.output_dir, "sub-dataset0", "checkpoint_temp.pth")
    ck = torch.load(ck_path, map_location="cpu", weights_only=False)
    assert ck["epoch"] == 0 and ck["optimizer"]["step"] == 8 and ck["optimizer"]["ema"] is not None
    # --resume --eval: the restored model is the saved one (evaluate is deterministic on the synthetic val split)
    a2 = parse(["--epochs", "2", "--output_dir", str(tmp_path / "b"), "--resume", ck_path])
    distill_sub.main(a2)
    assert a2.start_epoch == 1
    lines = [json.loads(l) for l in open(os.path.join(a2.output_dir, "sub-dataset0", "log.txt")).read().splitlines()]
    assert [l["epoch"] for l in lines] == [1] and lines[0]["train_loss"] == lines[0]["train_loss"]
    ck2 = torch.load(os.path.join(a2.output_dir, "sub-dataset0", "checkpoint_temp.pth"), map_location="cpu", weights_only=False)
    assert ck2["epoch"] == 1 and ck2["optimizer"]["step"] == 16
    moved = max(float((ck2["model"][k].float() - ck["model"][k].float()).abs().max()) for k in ck["model"])
    assert 0 < moved < 0.1                     # continued from the checkpoint, not from a fresh initialisation
    ema_gap = max(float((ck2["model_ema"][k] - ck["model_ema"][k]).abs().max()) for k in ck["model_ema"])
    assert ema_gap < 1e-3                      # EMA (decay 0.99996) restored, then moved by eight tiny updates


def test_distill_sub_finetune_and_eval(tmp_path, capsys):
    """--model-path/--finetune (distill_sub.py:205-226: a 1000-class checkpoint, then reset_classifier to the
    sub-dataset's classes) and --eval --resume (:390-393)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import argparse
    import devit_amd
    import distill_sub
    torch.manual_seed(5)
    pre = devit_amd.create_model("dedeit", num_classes=1000)
    torch.save({"model": pre.state_dict()}, tmp_path / "pre.pth")
    base = ["--synthetic", "2", "--batch-size", "4", "--model", "dedeit", "--teacher-model",
            "deit_base_distilled_patch16_224", "--dataset", "cifar100", "--num_division", "4", "--warmup-epochs", "0",
            "--epochs", "1"]
    parse = lambda extra: argparse.ArgumentParser(parents=[distill_sub.get_args_parser()]).parse_args(base + extra)
    a = parse(["--output_dir", str(tmp_path / "ft"), "--model-path", str(tmp_path / "pre.pth"), "--finetune"])
    distill_sub.main(a)
    ck = torch.load(os.path.join(a.output_dir, "sub-dataset0", "checkpoint_temp.pth"), map_location="cpu", weights_only=False)
    assert ck["model"]["head.weight"].shape == (25, 384) and ck["model"]["head_dist.weight"].shape == (25, 384)
    # two steps at lr ~4e-6 leave the backbone next to the pretrained weights (it was loaded, not re-initialised)
    k = "blocks.5.mlp.fc1.weight"
    assert float((ck["model"][k] - pre.state_dict()[k]).abs().max()) < 1e-3
    capsys.readouterr()
    e = parse(["--output_dir", str(tmp_path / "ev"), "--resume", os.path.join(a.output_dir, "sub-dataset0", "checkpoint_temp.pth"), "--eval"])
    distill_sub.main(e)
    printed = capsys.readouterr().out
    assert "acc1" in printed and not os.path.exists(os.path.join(e.output_dir, "sub-dataset0", "log.txt"))


def test_ensemble_cli(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import argparse
    import ensemble
    parser = argparse.ArgumentParser(parents=[ensemble.get_args_parser()], conflict_handler='resolve')
    args = parser.parse_args(["--synthetic", "2", "--batch-size", "4", "--epochs", "1", "--model", "dedeit",
                              "--teacher-model", "deit_base_distilled_patch16_224", "--output_dir", str(tmp_path)])
    args.output_dir = str(tmp_path)
    ensemble.main(args)
    assert os.path.exists(os.path.join(str(tmp_path), "checkpoint_temp.pth"))
    import json
    line = json.loads(open(os.path.join(str(tmp_path), "log.txt")).read().splitlines()[-1])
    assert line["train_loss"] == line["train_loss"] and "test_acc1" in line


@pytest.mark.parametrize("extra,expect", [([], "SoftTargetCrossEntropy"),
                                          (["--mixup", "0", "--cutmix", "0"], "LabelSmoothingCrossEntropy"),
                                          (["--mixup", "0", "--cutmix", "0", "--smoothing", "0"], "CrossEntropyLoss")])
def test_ensemble_base_criterion_selection(tmp_path, monkeypatch, extra, expect):
    """ensemble.py:350-357 picks the base criterion the way distill_sub.py:345-352 does: SoftTargetCrossEntropy with mixup,
    LabelSmoothingCrossEntropy(--smoothing, default 0.1) without it, nn.CrossEntropyLoss with neither."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import argparse
    import json
    import ensemble
    from devit_amd import losses
    seen = []
    real = losses.EnsLoss

    class Rec(real):
        def __init__(self, base, *a, **k):
            seen.append(base)
            super().__init__(base, *a, **k)
    monkeypatch.setattr(losses, "EnsLoss", Rec)
    parser = argparse.ArgumentParser(parents=[ensemble.get_args_parser()], conflict_handler='resolve')
    args = parser.parse_args(["--synthetic", "2", "--batch-size", "4", "--epochs", "1", "--model", "dedeit",
                              "--teacher-model", "deit_base_distilled_patch16_224", "--output_dir", str(tmp_path)] + extra)
    args.output_dir = str(tmp_path)
    ensemble.main(args)
    assert type(seen[0]).__name__ == expect
    if expect == "LabelSmoothingCrossEntropy":
        assert seen[0].smoothing == 0.1
    line = json.loads(open(os.path.join(str(tmp_path), "log.txt")).read().splitlines()[-1])
    assert line["train_loss"] == line["train_loss"]


@pytest.mark.parametrize("extra", [[], ["--distillation-type", "hard"], ["--mixup", "0", "--cutmix", "0", "--distillation-type", "soft"]])
def test_train_subdata_cli(tmp_path, extra):
    """train_subdata.py (teacher / fine-tuning loop, DistillationLoss with the teacher inside the criterion)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import argparse
    import train_subdata
    parser = argparse.ArgumentParser(parents=[train_subdata.get_args_parser()], conflict_handler='resolve')
    args = parser.parse_args(["--synthetic", "3", "--batch-size", "4", "--epochs", "1", "--model", "dedeit",
                              "--teacher-model", "deit_base_distilled_patch16_224", "--dataset", "cifar100",
                              "--num_division", "4", "--output_dir", str(tmp_path), "--warmup-epochs", "0"] + extra)
    train_subdata.main(args)
    out = os.path.join(str(tmp_path), "sub-dataset0")
    import json
    line = json.loads(open(os.path.join(out, "log.txt")).read().splitlines()[-1])
    assert line["train_loss"] == line["train_loss"] and "test_acc1" in line
    ck = torch.load(os.path.join(out, "checkpoint_temp.pth"), map_location="cpu", weights_only=False)
    assert len(ck["model"]) == 155


def test_distillation_loss_equals_distill_loss():
    """losses.DistillationLoss (teacher inside, utils/losses.py:44-118) == DistillLoss on the teacher's logits."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from devit_amd import losses
    dev = torch.device("cuda")
    torch.manual_seed(0)
    B, C = 8, 25
    lo, lk, lt = (torch.randn(B, C, device=dev, requires_grad=r) for r in (True, True, False))
    y = torch.randint(0, C, (B,), device=dev)

    class T(torch.nn.Module):
        def forward(self, x):
            return lt
    for kind in ("hard", "soft", "none"):
        for base, ref_base in ((torch.nn.CrossEntropyLoss(), torch.nn.CrossEntropyLoss()),
                               (losses.LabelSmoothingCrossEntropy(0.1), torch.nn.CrossEntropyLoss(label_smoothing=0.1))):
            got = losses.DistillationLoss(base, T(), kind, 0.5, 2.0)(inputs=None, outputs=(lo, lk), labels=y)
            b = ref_base(lo, y)
            if kind == "none":
                ref = b
            elif kind == "hard":
                ref = 0.5 * b + 0.5 * torch.nn.functional.cross_entropy(lk, lt.argmax(1))
            else:
                ref = 0.5 * b + 0.5 * torch.nn.functional.kl_div(torch.log_softmax(lk / 2, 1), torch.log_softmax(lt / 2, 1),
                                                                 reduction="sum", log_target=True) * 4 / lk.numel()
            got, ref = got.detach(), ref.detach()
            assert abs(float(got) - float(ref)) < 1e-5 * abs(float(ref)), (kind, float(got), float(ref))


def test_ensemble_cli_checkpoints_gates_shrink(tmp_path, capsys):
    """ensemble.py --eval from files: four sub-model checkpoints (`{model-path}/sub-dataset{i}/checkpoint.pth`), the
    teacher checkpoint, gates persisted beside them, and the physically shrunk model giving the masked model's scores."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import argparse
    import re
    import devit_amd
    import ensemble
    from devit_amd import shrink
    from devit_amd.ensemble_models import MultiViT
    torch.manual_seed(11)
    for i in range(4):
        os.makedirs(tmp_path / "subs" / f"sub-dataset{i}")
        torch.save(devit_amd.create_model("dedeit", num_classes=25).state_dict(),
                   tmp_path / "subs" / f"sub-dataset{i}" / "checkpoint.pth")
    torch.save(devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=100).state_dict(), tmp_path / "teacher.pth")
    multi = MultiViT(model="dedeit", num_div=4, num_classes_list=[25] * 4)
    g = torch.Generator().manual_seed(3)
    pol = []
    for _ in range(4 * 12):                      # per block: 4 of 6 heads, 1024 of 1536 neurons stay
        h, n = torch.zeros(6), torch.zeros(1536)
        h[torch.randperm(6, generator=g)[:4]] = 1
        n[torch.randperm(1536, generator=g)[:1024]] = 1
        pol.append((h, n))
    shrink.load_policy(multi, pol)
    shrink.save_gates(multi, tmp_path / "gates.pt")
    base = ["--synthetic", "8", "--batch-size", "4", "--model", "dedeit", "--teacher-model", "deit_base_distilled_patch16_224",
            "--model-path", str(tmp_path / "subs"), "--teacher-path", str(tmp_path / "teacher.pth"), "--eval",
            "--gates", str(tmp_path / "gates.pt"), "--output_dir", str(tmp_path)]
    parse = lambda extra: argparse.ArgumentParser(parents=[ensemble.get_args_parser()], conflict_handler='resolve').parse_args(base + extra)
    scores = []
    for extra in ([], ["--physical-shrink"]):
        capsys.readouterr()
        ensemble.main(parse(extra))
        out = capsys.readouterr().out
        m = re.search(r"\{'loss': ([0-9.eE+-]+), 'acc1': ([0-9.eE+-]+), 'acc5': ([0-9.eE+-]+)\}", out)
        assert m, out[-400:]
        scores.append([float(v) for v in m.groups()])
        if extra:
            assert "physically shrunk 48 blocks" in out
    assert abs(scores[0][0] - scores[1][0]) < 2e-2 * abs(scores[0][0]) and scores[0][1:] == scores[1][1:]


def test_distill_sub_to_ensemble_chain(tmp_path):
    """distill_sub.py writes `<out>/sub-dataset{k}/checkpoint.pth` (distill_sub.py:403-404,446-449) and ensemble.py reads
    `{--model-path}/sub-dataset{i}/checkpoint.pth` (ensemble.py:228): four divisions trained by the first CLI feed the
    second with NO file moved in between."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import argparse
    import json
    import distill_sub
    import ensemble
    outs = set()
    for k in range(4):
        a = argparse.ArgumentParser(parents=[distill_sub.get_args_parser()]).parse_args(
            ["--synthetic", "2", "--batch-size", "4", "--epochs", "1", "--model", "dedeit", "--teacher-model",
             "deit_base_distilled_patch16_224", "--dataset", "cifar100", "--num_division", "4", "--start-division", str(k),
             "--output_dir", str(tmp_path / "distill"), "--warmup-epochs", "0"])
        distill_sub.main(a)
        outs.add(a.output_dir)
        # eval accuracy on random labels can be 0 and then no best checkpoint is written (max_accuracy < acc1 is strict,
        # distill_sub.py:445): the chain needs the file, so fall back to the epoch checkpoint's model like a user would
        best = os.path.join(a.output_dir, f"sub-dataset{k}", "checkpoint.pth")
        if not os.path.exists(best):
            ck = torch.load(os.path.join(a.output_dir, f"sub-dataset{k}", "checkpoint_temp.pth"), map_location="cpu", weights_only=False)
            torch.save(ck["model"], best)
    assert len(outs) == 1                                  # one run directory, four sub-dataset{k} children
    model_path = outs.pop()
    e = argparse.ArgumentParser(parents=[ensemble.get_args_parser()], conflict_handler='resolve').parse_args(
        ["--synthetic", "2", "--batch-size", "4", "--epochs", "1", "--model", "dedeit", "--teacher-model",
         "deit_base_distilled_patch16_224", "--model-path", model_path, "--output_dir", str(tmp_path / "ens")])
    assert e.clip_grad is None and e.weight_decay == 0.05 and e.epochs == 1 and e.lr == 1e-5     # ensemble.py:41,68,72,77
    assert argparse.ArgumentParser(parents=[ensemble.get_args_parser()], conflict_handler='resolve').parse_args([]).epochs == 3
    e.output_dir = str(tmp_path / "ens")
    os.makedirs(e.output_dir, exist_ok=True)
    ensemble.main(e)
    line = json.loads(open(os.path.join(e.output_dir, "log.txt")).read().splitlines()[-1])
    assert line["train_loss"] == line["train_loss"]
    ck = torch.load(os.path.join(e.output_dir, "checkpoint_temp.pth"), map_location="cpu", weights_only=False)
    assert {"model", "ens_model", "optimizer", "ens_optimizer", "lr_scheduler", "ens_lr_scheduler", "epoch", "scaler", "args"} <= set(ck)
    # the backbones really came from the four distill_sub checkpoints (positional copy), then trained one tiny epoch
    sub0 = torch.load(os.path.join(model_path, "sub-dataset0", "checkpoint.pth"), map_location="cpu")
    k0 = [k for k in ck["model"] if k.endswith("blocks.0.attn.qkv.weight")][0]
    assert float((ck["model"][k0] - sub0["blocks.0.attn.qkv.weight"]).abs().max()) < 1e-3


def test_distill_sub_shrink_flags(tmp_path):
    """--shrink_checkpoint / --neuron_shrinking / --head_shrinking (distill_sub.py:383-401): the policy files are read,
    one batch is ranked, the student trains GATED (never silently dense), the gates are persisted; a shrink flag without
    a policy directory is an error; unsupported --opt / --sched values are refused."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import argparse
    import numpy as np
    import distill_sub
    from devit_amd import shrink
    pol = np.zeros((3, 24))
    pol[1, :12], pol[1, 12:] = 0.25, 0.34            # best row: 25 % of the neurons, int(6 * .66) = 3 heads kept -> 3 off
    pol[0], pol[2] = 0.5, 0.1
    os.makedirs(tmp_path / "shrink")
    np.save(tmp_path / "shrink" / "shrinked_policy.npy", pol)
    np.save(tmp_path / "shrink" / "shrinked_accuracy.npy", np.array([10.0, 80.0, 30.0]))
    ns, hs = shrink.read_shrink_checkpoint(str(tmp_path / "shrink"))
    assert np.allclose(ns, 0.25) and np.allclose(hs, 0.34) and len(hs) == 12
    np.save(tmp_path / "shrink" / "p25.npy", np.concatenate([pol, np.ones((3, 1))], 1))      # 25-column files work too
    base = ["--synthetic", "2", "--batch-size", "4", "--epochs", "1", "--model", "dedeit", "--teacher-model",
            "deit_base_distilled_patch16_224", "--dataset", "cifar100", "--num_division", "4", "--warmup-epochs", "0"]
    parse = lambda extra: argparse.ArgumentParser(parents=[distill_sub.get_args_parser()]).parse_args(base + extra)
    a = parse(["--output_dir", str(tmp_path / "o"), "--shrink_checkpoint", str(tmp_path / "shrink"), "--neuron_shrinking",
               "--head_shrinking"])
    distill_sub.main(a)
    gates = torch.load(os.path.join(a.output_dir, "sub-dataset0", "gates.pt"), weights_only=False) \
        if os.path.exists(os.path.join(a.output_dir, "sub-dataset0", "gates.pt")) else None
    if gates is not None:                             # written with the best checkpoint (needs acc1 > 0 on random labels)
        assert all(int(h.sum()) == 3 and int(n.sum()) == 1152 for h, n in gates)
    # the default trains through the compacted blocks; --no-physical-shrink trains the masked model at the dense cost, as the
    # reference does: same seed, same data, same gates -> the same epoch within bf16 noise
    import json
    b = parse(["--output_dir", str(tmp_path / "m"), "--shrink_checkpoint", str(tmp_path / "shrink"), "--neuron_shrinking",
               "--head_shrinking", "--no-physical-shrink"])
    distill_sub.main(b)
    la, lb = (json.loads(open(os.path.join(x.output_dir, "sub-dataset0", "log.txt")).read().splitlines()[-1]) for x in (a, b))
    assert abs(la["train_loss"] - lb["train_loss"]) < 2e-2 * abs(lb["train_loss"]), (la["train_loss"], lb["train_loss"])
    with pytest.raises(ValueError):
        distill_sub.main(parse(["--output_dir", str(tmp_path / "p"), "--neuron_shrinking"]))
    for bad in (["--opt", "sgd"], ["--sched", "step"]):
        with pytest.raises(SystemExit):
            distill_sub.main(parse(["--output_dir", str(tmp_path / "q")] + bad))


def test_rank_units_and_masks(tmp_path):
    """shrink.rank_units -> masks_from_sparsity: the kept units are the highest-ranked ones and the gated forward's
    neuron_output / head_output are zero exactly on the masked units (the contract core/imp_rank.py relies on)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import devit_amd
    from devit_amd import shrink
    dev = torch.device("cuda")
    torch.manual_seed(2)
    m = devit_amd.create_model("dedeit", num_classes=25).to(dev)
    loader = [(torch.randn(6, 3, 224, 224, device=dev), torch.zeros(6, dtype=torch.long, device=dev))]
    nr, hr = shrink.rank_units(m, loader, dev)
    assert len(nr) == 12 and len(hr) == 12 and sorted(nr[0].tolist()) == list(range(1536)) and sorted(hr[3].tolist()) == list(range(6))
    # the activation-mass term dominates the neuron score (0.9 weight): the top-ranked neuron carries far more |activation|
    # than the bottom-ranked one
    act = m.blocks[0].mlp.neuron_output.float().abs().sum((0, 1))
    assert float(act[nr[0][-1]]) > float(act[nr[0][0]])
    pol = shrink.masks_from_sparsity(m, [0.5] * 12, [0.34] * 12, nr, hr)
    shrink.load_policy(m, pol)
    m.eval()
    with torch.no_grad():
        m(loader[0][0])
    for i in (0, 7):
        hm, nm = pol[i]
        assert int(hm.sum()) == 3 and int(nm.sum()) == 768
        assert set(torch.nonzero(nm).reshape(-1).tolist()) == set(nr[i][::-1][:768].tolist())
        no = m.blocks[i].mlp.neuron_output.float().abs().sum((0, 1)).cpu()
        ho = m.blocks[i].attn.head_output.float().abs().sum((0, 1, 3)).cpu()
        assert bool((no[nm == 0] == 0).all()) and bool((no[nm == 1] > 0).all())
        assert bool((ho[hm == 0] == 0).all()) and bool((ho[hm == 1] > 0).all())


@pytest.mark.parametrize("extra,steps", [(["--no-repeated-aug"], 2), ([], 8)])
def test_distill_sub_on_a_dataset_provider(tmp_path, monkeypatch, extra, steps):
    """Without --synthetic the CLI takes its datasets from `data.get_dataset.build_division_dataset` (the reference's
    package; a stand-in with host tensors here), builds the reference's samplers / loaders (distill_sub.py:269-313) and
    trains and evaluates from pinned host batches: one epoch, the reference's files written, `steps` iterations
    (--no-repeated-aug: the train sampler runs over the 8-image TEST set as in the reference; default: RASampler, 256 of the
    3 x 260 repeats at batch 32)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import argparse
    import json
    import types
    import distill_sub
    pkg, mod = types.ModuleType("data"), types.ModuleType("data.get_dataset")

    def tensors(n, seed):
        g = torch.Generator().manual_seed(seed)
        return torch.utils.data.TensorDataset(torch.randn((n, 3, 224, 224), generator=g), torch.randint(0, 25, (n,), generator=g))
    seen = []

    def build_division_dataset(dataset_path, args):
        seen.append(dataset_path)
        return tensors(260, 1), tensors(8, 2), 25
    mod.build_division_dataset = build_division_dataset
    pkg.get_dataset = mod
    monkeypatch.setitem(sys.modules, "data", pkg)
    monkeypatch.setitem(sys.modules, "data.get_dataset", mod)
    from devit_amd import engine
    counted = []
    real = engine.distill_forward

    def counting(*a, **k):
        counted.append(1)
        return real(*a, **k)
    monkeypatch.setattr(engine, "distill_forward", counting)
    parser = argparse.ArgumentParser(parents=[distill_sub.get_args_parser()])
    bs = "4" if extra else "32"
    args = parser.parse_args(["--batch-size", bs, "--eval-batch-size", "8", "--epochs", "1", "--model", "dedeit", "--num_workers", "0",
                              "--teacher-model", "deit_base_distilled_patch16_224", "--dataset", "cifar100", "--num_division", "4",
                              "--data-path", str(tmp_path / "data"), "--output_dir", str(tmp_path / "out"), "--warmup-epochs", "0",
                              "--synthetic", "0", "--teacher-path", ""] + extra)
    try:
        distill_sub.main(args)
    except SystemExit as e:                      # no teacher checkpoint on a real-data run is an error by design
        assert "teacher checkpoint" in str(e)
        args.teacher_path = str(tmp_path / "teachers")
        import devit_amd
        os.makedirs(os.path.join(args.teacher_path, "sub-dataset0"))
        torch.save(devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=25).state_dict(),
                   os.path.join(args.teacher_path, "sub-dataset0", "checkpoint.pth"))
        counted.clear()
        distill_sub.main(args)
    assert seen and seen[-1] == os.path.join(str(tmp_path / "data"), "sub-dataset0")
    assert len(counted) == steps
    out = os.path.join(args.output_dir, "sub-dataset0")
    line = json.loads(open(os.path.join(out, "log.txt")).read().splitlines()[-1])
    assert line["train_loss"] == line["train_loss"] and 0.0 <= line["test_acc1"] <= 100.0
