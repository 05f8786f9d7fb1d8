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
    out = args.output_dir
    assert os.path.exists(os.path.join(out, "checkpoint_temp.pth")) and os.path.exists(os.path.join(out, "log.txt"))
    ck = torch.load(os.path.join(out, "checkpoint_temp.pth"), map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "epoch", "model_ema", "scaler", "args"}
    assert len(ck["model"]) == 155 and len(ck["model_ema"]) == 155
    import json
    line = json.loads(open(os.path.join(out, "log.txt")).read().splitlines()[-1])
    assert line["train_loss"] == line["train_loss"] and line["n_parameters"] == 21685682   # finite, C = 25


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
