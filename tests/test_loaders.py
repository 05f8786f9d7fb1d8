"""Host-side data plumbing of the re-hosted CLIs (distill_sub.build_loaders): the repeated-augmentation sampler against the
reference's own (utils/samplers.py:8-63, golden written by tests/golden/make_golden.py from the imported reference) and the
sampler / loader wiring of distill_sub.py:269-313, ensemble.py:261-300 on a stand-in `data.get_dataset` provider."""
import argparse
import json
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Len:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def test_ra_sampler_vs_reference_golden():
    import distill_sub
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "ra_sampler.json")))
    assert len(cases) == 18
    for c in cases:
        s = distill_sub.RASampler(_Len(c["n"]), num_replicas=c["world"], rank=c["rank"], shuffle=True)
        s.set_epoch(c["epoch"])
        idx = list(s)
        assert len(s) == c["length"] == len(idx), c
        assert idx[:24] == c["head"], c
        assert sum((i + 1) * (v + 1) for i, v in enumerate(idx)) % 1000000007 == c["checksum"], c


def _provider(monkeypatch, n_train, n_test, calls):
    pkg, mod = types.ModuleType("data"), types.ModuleType("data.get_dataset")

    def tensors(n, seed):
        g = torch.Generator().manual_seed(seed)
        return torch.utils.data.TensorDataset(torch.randn((n, 3, 8, 8), generator=g), torch.randint(0, 25, (n,), generator=g))

    def build_division_dataset(dataset_path, args):
        calls.append(("division", dataset_path))
        return tensors(n_train, 1), tensors(n_test, 2), 25

    def build_dataset(args):
        calls.append(("whole", args.data_path))
        return tensors(n_train, 3), tensors(n_test, 4), 100
    mod.build_division_dataset, mod.build_dataset = build_division_dataset, build_dataset
    pkg.get_dataset = mod
    monkeypatch.setitem(sys.modules, "data", pkg)
    monkeypatch.setitem(sys.modules, "data.get_dataset", mod)


def _args(extra=()):
    import distill_sub
    p = argparse.ArgumentParser(parents=[distill_sub.get_args_parser()])
    return p.parse_args(["--batch-size", "16", "--eval-batch-size", "10", "--num_workers", "0", "--data-path", "/d",
                         "--start-division", "2"] + list(extra))


def test_build_loaders_division_provider(monkeypatch):
    import distill_sub
    calls = []
    _provider(monkeypatch, 600, 95, calls)
    tr, va, nc = distill_sub.build_loaders(_args(), 7, "cpu", provider="division")
    assert calls == [("division", "/d/sub-dataset2")] and nc == 25                 # distill_sub.py:269-272
    assert isinstance(tr.sampler, distill_sub.RASampler) and tr.drop_last and not va.drop_last
    assert len(tr.sampler) == 600 // 256 * 256 and len(tr) == 512 // 16            # an epoch is 512 draws of the 1800 repeats
    assert isinstance(va.sampler, torch.utils.data.SequentialSampler) and len(va) == 10 and va.batch_size == 10
    x, y = next(iter(tr))
    assert x.shape == (16, 3, 8, 8) and y.dtype == torch.int64
    # --no-repeated-aug: distill_sub.py:281-283 builds the train sampler over the TEST set; kept (epoch length)
    tr2, _, _ = distill_sub.build_loaders(_args(["--no-repeated-aug"]), 7, "cpu", provider="division")
    assert isinstance(tr2.sampler, torch.utils.data.DistributedSampler) and len(tr2.sampler) == 95 and len(tr2) == 95 // 16
    # --dist-eval
    _, va3, _ = distill_sub.build_loaders(_args(["--dist-eval"]), 7, "cpu", provider="division")
    assert isinstance(va3.sampler, torch.utils.data.DistributedSampler) and not va3.sampler.shuffle
    distill_sub.set_epoch(tr, 5)
    assert tr.sampler.epoch == 5


def test_build_loaders_whole_provider(monkeypatch):
    import distill_sub
    calls = []
    _provider(monkeypatch, 300, 40, calls)
    tr, va, nc = distill_sub.build_loaders(_args(["--no-repeated-aug"]), 100, "cpu", provider="whole", plain_sampler_over="train")
    assert calls == [("whole", "/d")] and nc == 100
    assert len(tr.sampler) == 300 and len(va) == 4                                 # ensemble.py:271-273: over the train set


def test_build_loaders_without_provider(monkeypatch):
    import distill_sub
    monkeypatch.setitem(sys.modules, "data", None)
    monkeypatch.setitem(sys.modules, "data.get_dataset", None)
    with pytest.raises(SystemExit) as e:
        distill_sub.build_loaders(_args(), 25, "cpu")
    assert "--synthetic" in str(e.value)
