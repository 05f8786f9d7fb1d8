"""CPU-only tests of the host side that mirrors the reference's module / registry / checkpoint surface."""
import json
import os

import pytest
import torch

import devit_amd
from devit_amd import de_vit, ddp, ops, registry

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_registry_semantics():
    assert {"dedeit", "devit", "deit_base_distilled_patch16_224", "deit_tiny_patch16_224",
            "vit_large_patch16_224"} <= set(registry.list_models())
    with pytest.raises(RuntimeError):
        registry.create_model("no_such_model")
    m = registry.create_model("dedeit", pretrained=False, num_classes=25, resize_dim=None, drop_rate=0.0,
                              drop_path_rate=0.1, drop_block_rate=None)        # None kwargs dropped (timm)
    assert m.num_classes == 25 and m.embed_dim == 384 and len(m.blocks) == 12
    assert de_vit.model_config["dedeit"]["embed_dim"] == 384
    assert de_vit.model_config["deit_base_distilled_patch16_224"]["embed_dim"] == 768
    # every registered name constructs; the D = 192 names of models/deit_vit.py:457-525 do not fit the MFMA kernels' tiles and are pinned to the
    # exact-fp32 kernels (de_vit.check_geometry); anything that is not 64-wide heads x a multiple of 64 refuses AT CONSTRUCTION with the reason
    for name in ("deit_tiny_patch16_224", "deit_tiny_distilled_patch16_224", "vit_tiny_patch16_224"):
        m = registry.create_model(name, num_classes=10)
        assert m.embed_dim == 192 and m.blocks[0].attn.num_heads == 3 and m.precision == "f32"
        with pytest.raises(Exception, match="pinned"):
            m.precision = "bf16"
        with pytest.warns(UserWarning, match="ignored"):          # what the CLIs do with --teacher-precision (default bf16) on such a teacher
            assert m.request_precision("bf16") == "f32" and m.precision == "f32"
    assert registry.create_model("dedeit", num_classes=10).precision == "bf16"
    assert registry.create_model("dedeit", num_classes=10).request_precision("f16") == "f16"
    assert registry.create_model("dedeit", embed_dim=320, num_heads=5, num_classes=10).precision == "f32"      # (5 heads x 64: narrow, fp32 kernels)
    with pytest.raises(NotImplementedError, match="multiples of 64"):
        registry.create_model("dedeit", embed_dim=320, num_heads=4)                                            # heads are not 64 wide
    with pytest.raises(NotImplementedError):
        registry.create_model("dedeit", embed_dim=160, num_heads=2)


def test_statedict_abi():
    c = json.load(open(os.path.join(GOLD, "statedict_keys.json")))
    s = devit_amd.create_model("dedeit", num_classes=25)
    t = devit_amd.create_model("deit_base_distilled_patch16_224", num_classes=25)
    assert [[k, list(v.shape)] for k, v in s.state_dict().items()] == c["dedeit_keys"]
    assert [[k, list(v.shape)] for k, v in t.state_dict().items()] == c["deitb_keys"]
    assert len(list(s.buffers())) == 0                                             # no buffers (ensemble.py:229-238)
    assert sorted(s.no_weight_decay()) == c["no_weight_decay"]
    n_mlp = sum(1 for m in s.modules() if 'Mlp' in str(m) and 'Attention' not in str(m))
    n_att = sum(1 for m in s.modules() if 'Attention' in str(m) and 'Mlp' not in str(m))
    assert (n_mlp, n_att) == (c["n_mlp"], c["n_attn"]) == (12, 12)               # core/imp_rank.py:30,107
    # positional copy as ensemble.py:192-200 does it: last four keys are the heads
    assert list(s.state_dict())[-4:] == ["head.weight", "head.bias", "head_dist.weight", "head_dist.bias"]


def test_gate_contract_and_drop_path_schedule():
    m = devit_amd.create_model("dedeit", num_classes=10, drop_path_rate=0.1)
    mlp, att = m.blocks[3].mlp, m.blocks[3].attn
    assert mlp.hidden_features == 1536 and att.num_heads == 6
    assert torch.equal(mlp.gate, torch.ones(1536)) and mlp.gate_on(torch.device("cpu")) is None
    g = torch.ones(1536); g[::3] = 0
    mlp.gate = g                                                                    # plain attribute assignment
    assert mlp.gate is g and torch.equal(mlp.gate_on(torch.device("cpu")), g)
    assert "gate" not in m.state_dict() and not any("gate" in k for k in m.state_dict())
    dpr = [b.drop_prob for b in m.blocks]
    assert dpr[0] == 0.0 and abs(dpr[-1] - 0.1) < 1e-7 and dpr == sorted(dpr)      # de_vit.py:175
    m.train()
    bps = [b.block_params.__func__ for b in m.blocks]                              # exists
    assert len(bps) == 12
    m.reset_classifier(7)
    assert m.head.out_features == 7 and m.head_dist.out_features == 7


def test_row_padding_and_splitk():
    assert ops.pad_rows(50688) == 50688 and ops.pad_rows(396) == 512 and ops.pad_rows(1) == 256
    assert ops.split_k_for(1536, 384, 792) == 14        # 36 tiles of 128x128 -> 504 workgroups (two per CU)
    assert ops.split_k_for(1152, 384, 792) == 18        # 27 tiles of 128x128, two resident per CU -> 486
    assert 1 <= ops.split_k_for(384, 384, 792) <= 792
    t = ops.rows_alloc(5, 8, torch.float32, torch.device("cpu"))
    assert t.shape == (256, 8) and bool((t[5:] == 0).all())


def test_flat_params_and_buckets():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Linear(32, 8), torch.nn.LayerNorm(8))
    ref = {n: p.detach().clone() for n, p in m.named_parameters()}
    flat = ddp.FlatParams(m)
    assert flat.names[0] == "2.bias" and flat.names[-1] == "0.weight"              # reverse forward order
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), ref[n]) and p.grad is not None and float(p.grad.abs().sum()) == 0
        assert p.data.data_ptr() >= flat.flat.data_ptr() and p.data_ptr() % 16 == 0
    red = ddp.BucketedGradReducer(flat, bucket_bytes=1024)
    covered = sorted(i for b in red.buckets for i in range(b[2], b[3] + 1))
    assert covered == list(range(len(flat.params)))
    assert red.buckets[0][0] == 0 and red.buckets[-1][1] == flat.numel
    fired = []
    red._launch = lambda b: fired.append(b) or red.launched.__setitem__(b, True)
    for p in flat.params:                                                          # backward order
        red.mark_ready([p])
    assert fired == list(range(len(red.buckets)))                                  # buckets fire in order


def test_prepared_batches_lookahead_order():
    """engine._PreparedBatches: every batch is yielded once, in order, after transfer + mixup, together with the
    teacher output that was submitted for exactly that batch one iteration earlier."""
    import torch
    from devit_amd import engine

    class FakeLook:
        def __init__(self):
            self.pending, self.log = None, []

        def submit(self, samples, defer=False):
            assert self.pending is None
            self.pending = samples
            self.log.append(("submit", float(samples[0])) + (("deferred",) if defer else ()))

        def launch(self):
            self.log.append(("launch", float(self.pending[0])))

        def take(self, samples):
            assert self.pending is samples
            self.pending = None
            self.log.append(("take", float(samples[0])))
            return {"output": samples * 10}

    loader = [(torch.tensor([float(i)]), torch.tensor([i])) for i in range(4)]
    mix = lambda s, t: (s + 0.5, t)
    look = FakeLook()
    got = list(engine._PreparedBatches(loader, "cpu", mix, look))
    assert [float(s[0]) for s, _, _ in got] == [0.5, 1.5, 2.5, 3.5]
    assert [float(o["output"][0]) for _, _, o in got] == [5.0, 15.0, 25.0, 35.0]
    # the first batch's teacher forward is enqueued at once; every later one is recorded and launched by launch_teacher() (the loop calls it behind the
    # student's forward: distill_forward(after_student=...)) or, at the latest, by the take()
    assert look.log[:4] == [("submit", 0.5), ("take", 0.5), ("submit", 1.5, "deferred"), ("take", 1.5)] and look.pending is None
    look2 = FakeLook()
    batches = engine._PreparedBatches(loader, "cpu", mix, look2)
    for _ in batches:
        batches.launch_teacher()
        break
    assert look2.log == [("submit", 0.5), ("take", 0.5), ("submit", 1.5, "deferred"), ("launch", 1.5)]
    # the real look-ahead object: a deferred submit launches at launch() or at take(), never twice
    calls = []
    real = engine.TeacherLookahead.__new__(engine.TeacherLookahead)
    real.teacher, real._pending = None, None
    orig = engine._teacher_forward_async
    engine._teacher_forward_async = lambda teacher, samples: (calls.append(float(samples[0])), (lambda: {"output": samples}))[1]
    try:
        a, b = torch.tensor([1.0]), torch.tensor([2.0])
        real.submit(a, defer=True)
        assert calls == []
        real.launch(); real.launch()
        assert calls == [1.0] and real.take(a)["output"] is a
        real.submit(b, defer=True)
        assert real.take(b)["output"] is b and calls == [1.0, 2.0]
        with pytest.raises(RuntimeError):
            real.take(b)
    finally:
        engine._teacher_forward_async = orig
    assert len(engine._PreparedBatches(loader, "cpu", None, None)) == 4
    assert [o for _, _, o in engine._PreparedBatches(loader, "cpu", None, None)] == [None] * 4
    assert list(engine._PreparedBatches([], "cpu", None, look)) == []


def test_compact_block_weights_equal_masked_block():
    """shrink.compact_block_weights: the compacted weights compute exactly the masked block (models/de_vit.py:35-47,
    65-87 with gates), zero-padded units included; non-binary gates are folded into proj / fc2."""
    import torch
    import torch.nn.functional as F
    from devit_amd import de_vit, shrink
    from oracle import devit_oracle as O

    torch.manual_seed(0)
    D, H = 384, 6
    blk = de_vit.Block(D, H, mlp_ratio=4., qkv_bias=True)
    for p in blk.parameters():
        torch.nn.init.normal_(p, std=0.05)
    hg = torch.tensor([1., 0., 0.5, 1., 0., 1.])                  # 4 kept -> 4 run; one real-valued gate
    ng = (torch.rand(1536) > 0.3).float()
    ng[5] = 0.25
    blk.attn.gate, blk.mlp.gate = hg, ng
    w = shrink.compact_block_weights(blk)
    assert w["num_heads"] == 4 and w["qkv_w"].shape == (3 * 256, D) and w["proj_w"].shape == (D, 256)
    assert w["fc1_w"].shape[0] % 128 == 0 and w["fc1_w"].shape[0] >= int((ng != 0).sum()) > w["fc1_w"].shape[0] - 128
    st = {"a.qkv.weight": blk.attn.qkv.weight, "a.qkv.bias": blk.attn.qkv.bias, "a.proj.weight": blk.attn.proj.weight,
          "a.proj.bias": blk.attn.proj.bias, "m.fc1.weight": blk.mlp.fc1.weight, "m.fc1.bias": blk.mlp.fc1.bias,
          "m.fc2.weight": blk.mlp.fc2.weight, "m.fc2.bias": blk.mlp.fc2.bias}
    x = torch.randn(2, 50, D)
    with torch.no_grad():
        ref_a, _, _ = O.attention(st, "a.", x, H, head_gate=hg)
        ref_m, _ = O.mlp(st, "m.", x, neuron_gate=ng)
        Hr = w["num_heads"]
        qkv = F.linear(x, w["qkv_w"], w["qkv_b"]).reshape(2, 50, 3, Hr, 64).permute(2, 0, 3, 1, 4)
        a = ((qkv[0] @ qkv[1].transpose(-2, -1)) * 0.125).softmax(-1)
        got_a = F.linear((a @ qkv[2]).transpose(1, 2).reshape(2, 50, Hr * 64), w["proj_w"], blk.attn.proj.bias)
        got_m = F.linear(F.gelu(F.linear(x, w["fc1_w"], w["fc1_b"])), w["fc2_w"], blk.mlp.fc2.bias)
    assert float((got_a - ref_a).abs().max()) < 2e-5 * float(ref_a.abs().max())
    assert float((got_m - ref_m).abs().max()) < 2e-5 * float(ref_m.abs().max())
    # odd number of kept heads -> one all-zero head is run; nothing kept -> minimum sizes
    blk.attn.gate = torch.tensor([1., 1., 1., 0., 0., 0.])
    assert shrink.compact_block_weights(blk)["num_heads"] == 4
    blk.attn.gate, blk.mlp.gate = torch.zeros(6), torch.zeros(1536)
    w0 = shrink.compact_block_weights(blk)
    assert w0["num_heads"] == 2 and w0["fc1_w"].shape[0] == 128 and float(w0["qkv_w"].abs().max()) == 0.0
    # masks from sparsities (core/imp_rank.py:50-62,132-144)
    class M(torch.nn.Module):
        def __init__(s):
            super().__init__(); s.b = blk
    pol = shrink.masks_from_sparsity(M(), [0.3], [0.34], [list(range(1536))], [[3, 1, 0, 2, 5, 4]])
    assert int(pol[0][0].sum()) == int(6 * (1 - 0.34)) == 3 and pol[0][0].tolist() == [0., 0., 1., 0., 1., 1.]
    assert int(pol[0][1].sum()) == int(1536 * 0.7) and pol[0][1][-1] == 1 and pol[0][1][0] == 0


def test_gate_persistence_roundtrip(tmp_path):
    import torch
    from devit_amd import de_vit, shrink
    blk = de_vit.Block(384, 6, qkv_bias=True)
    holder = torch.nn.Sequential(blk)
    hm, nm = torch.tensor([1., 0., 1., 1., 0., 1.]), (torch.arange(1536) % 3 != 0).float()
    shrink.load_policy(holder, [(hm, nm)])
    shrink.save_gates(holder, tmp_path / "gates.pt")
    shrink.load_policy(holder, [(torch.ones(6), torch.ones(1536))])
    shrink.load_gates(holder, tmp_path / "gates.pt")
    assert torch.equal(blk.attn.gate, hm) and torch.equal(blk.mlp.gate, nm)
    assert "gate" not in "".join(holder.state_dict().keys())          # checkpoint ABI unchanged


# ------------------------------------------------------------------------------------------ round-2 known answers
def test_flop_accounting_matches_reference_formulas():
    """core/compute_metric.py:1-69 evaluated by tests/golden/make_golden.py (flops.json): the build's accounting
    (devit_amd.flops, used by bench.py's roofline and BASELINE.md section 2) reproduces it to the last digit."""
    import json
    import bench
    from devit_amd import flops as F
    g = json.load(open(os.path.join(GOLDEN, "flops.json")))
    s = dict(emb=384, seq_length=197, mlp_ratio=4, head=6, layer=12, num_class=1000)
    assert F.forward_gflops(**s) == g["dedeit_dense_gflops"] == 9.197764608
    assert F.params_m(**s) == g["dedeit_dense_mparams"] == 22.03684
    assert F.forward_gflops() == g["deitb_dense_gflops"] == 35.127656448 and F.params_m() == g["deitb_dense_mparams"]
    assert F.forward_gflops(**s, neuron_sparsity=[0.3] * 12, head_sparsity=[0.3] * 12) == g["dedeit_shrunk_0.3_gflops"]
    assert F.params_m(**s, neuron_sparsity=[0.3] * 12, head_sparsity=[0.3] * 12) == g["dedeit_shrunk_0.3_mparams"]
    mixed = dict(neuron_sparsity=[0.1 * (i % 4) for i in range(12)], head_sparsity=[0.17 * (i % 3) for i in range(12)])
    assert F.forward_gflops(**s, **mixed) == g["dedeit_mixed_gflops"]
    assert F.macs_g(**s) == g["dedeit_dense_gflops"] / 2
    # the benchmark's geometry (198 tokens, 25 classes) and the constant bench.py prices the step with
    assert F.forward_gflops(**{**s, "seq_length": 198, "num_class": 25}) == g["dedeit_c25_n198_gflops"]
    assert F.forward_gflops(seq_length=198, num_class=25) == g["deitb_c25_n198_gflops"]
    assert abs(F.step_gflops_per_image() - bench.GFLOP_PER_IMG_STEP) < 1e-3 and abs(bench.GFLOP_PER_IMG_STEP - 63.503) < 1e-9
    # physically shrunk models: the run sizes are padded to kernel granules (heads even, hidden % 128), so the FLOPs that
    # actually run sit between the analytic shrunk cost and the dense one
    import devit_amd
    from devit_amd import shrink
    m = devit_amd.create_model("dedeit", num_classes=1000)
    dense = shrink.compacted_gflops(m, tokens=197)
    assert abs(dense - g["dedeit_dense_gflops"]) < 0.01 * dense          # second head + exact conv vs the formula's 2*3*emb*224^2


def test_importance_ranking_matches_reference():
    """shrink.neuron_scores / head_scores / masks_from_sparsity against core/imp_rank.py's own mlp_neuron_rank,
    attn_head_rank, mlp_neuron_mask, attn_head_mask run on synthetic activations (tests/golden/imp_rank.npz)."""
    import numpy as np
    from devit_amd import shrink
    g = dict(np.load(os.path.join(GOLDEN, "imp_rank.npz")))
    prob = torch.softmax(torch.from_numpy(g["logits"]), -1)
    for i in range(2):
        nr = np.argsort(shrink.neuron_scores(torch.from_numpy(g[f"n{i}"]), prob).numpy())
        hr = np.argsort(shrink.head_scores(torch.from_numpy(g[f"h{i}"]), prob).numpy())
        assert np.array_equal(nr, g["neuron_rank"][i]) and np.array_equal(hr, g["head_rank"][i])

    class Blk:           # the two attributes masks_from_sparsity reads
        def __init__(self, H, hid):
            self.attn = type("A", (), {"num_heads": H})()
            self.mlp = type("M", (), {"hidden_features": hid})()
    real = shrink._blocks
    shrink._blocks = lambda model: model
    try:
        pol = shrink.masks_from_sparsity([Blk(4, 48), Blk(4, 48)], g["sparsity"], g["sparsity"], g["neuron_rank"], g["head_rank"])
    finally:
        shrink._blocks = real
    for i, (hm, nm) in enumerate(pol):
        assert np.array_equal(hm.numpy(), g["head_mask"][i]) and np.array_equal(nm.numpy(), g["neuron_mask"][i])


def test_bench_cites_the_profile_taken_on_the_current_kernel_sources(tmp_path, monkeypatch):
    """bench.committed_counter / parity_statement pick the summary whose kernel_sources_hash equals the current one --
    never the lexicographically last file name ('r99_C_*' sorts before 'r99_w_*': round 3's driver line cited the stale set
    and reported traffic / mfma_busy as null)."""
    import json
    import time
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    cur = bench.kernel_sources_hash()
    (prof / "r99_C_pmc_traffic.json").write_text(json.dumps({"kernel_sources_hash": cur, "traffic_bytes_per_launch": 111}))
    (prof / "r99_C_pmc_mfma.json").write_text(json.dumps({"kernel_sources_hash": cur, "mfma_busy": 0.5}))
    (prof / "r99_C_parity_margins.json").write_text(json.dumps({"kernel_sources_hash": cur, "rows": [
        {"test": "tests/test_gpu_model.py::test_model_forward_vs_golden[dedeit]", "line": 1, "value": 0.008, "bar": 1.5e-2}]}))
    time.sleep(0.02)       # the stale set is NEWER by mtime and LATER by name: the hash must still win
    (prof / "r99_w_pmc_traffic.json").write_text(json.dumps({"kernel_sources_hash": "0" * 16, "traffic_bytes_per_launch": 222}))
    (prof / "r99_w_pmc_mfma.json").write_text(json.dumps({"kernel_sources_hash": "0" * 16, "mfma_busy": 0.1}))
    (prof / "r99_w_parity_margins.json").write_text(json.dumps([
        {"test": "tests/test_gpu_model.py::test_model_forward_vs_golden[dedeit]", "line": 1, "value": 0.001, "bar": 1.5e-2}]))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_sources_hash", lambda: cur)
    v, src = bench.committed_counter("*_pmc_traffic.json", "traffic_bytes_per_launch")
    assert v == 111 and src["file"].endswith("r99_C_pmc_traffic.json") and src["stale"] is False
    v, src = bench.committed_counter("*_pmc_mfma.json", "mfma_busy")
    assert v == 0.5 and src["stale"] is False
    ps = bench.parity_statement()
    assert ps["logits_rel_to_max"] == {"dedeit": 0.008} and "r99_C_" in ps["source"] and ps["stale"] is False
    # nothing matches the current sources: the newest by mtime is cited and marked stale, its value withheld
    monkeypatch.setattr(bench, "kernel_sources_hash", lambda: "f" * 16)
    v, src = bench.committed_counter("*_pmc_traffic.json", "traffic_bytes_per_launch")
    assert v is None and src["stale"] is True and src["file"].endswith("r99_w_pmc_traffic.json")
    assert bench.parity_statement()["stale"] is True


def test_deferred_wgrad_groups(monkeypatch):
    """ops.DeferredWgrads: which blocks' weight gradients share one launch of the grouped kernel.  Without a gradient exchange to overlap: all blocks of
    the encoder call, launched behind block 0; with a reducer of world > 1 on grad_ready: the blocks of one bucket (so that every bucket still leaves as
    soon as its gradients exist); DEVIT_WGRAD_GROUP forces either; a full job table flushes early."""
    from devit_amd import ops, _lib as L

    class BP:
        def __init__(self, i):
            self.i = i

        def all_params(self):
            return [self.i]

        def finish_grads(self):
            pass

    class Hook:
        world = 2

        def __init__(self):
            self.seen = []

        def __call__(self, params):
            self.seen += params

        def bucket_of(self, params):
            return {10: 0, 9: 1, 8: 1, 7: 1, 6: 2, 5: 2, 4: 2, 3: 3, 2: 3, 1: 4, 0: 5}[params[0]]

    class Cfg:
        pass
    monkeypatch.delenv("DEVIT_WGRAD_GROUP", raising=False)
    cfg = Cfg()
    cfg.blocks = [BP(i) for i in range(11)]
    cfg.grad_ready = Hook()
    d = ops.DeferredWgrads(cfg, 11)
    assert d.policy == "bucket"
    assert [i for i in range(10, -1, -1) if d.last_of_group(i)] == [10, 7, 4, 2, 1, 0]
    cfg.grad_ready.world = 1                                    # nothing to overlap: one group
    d = ops.DeferredWgrads(cfg, 11)
    assert d.policy == "all" and [i for i in range(10, -1, -1) if d.last_of_group(i)] == [0]
    d.jobs = [None] * (L.WGRAD_MAX_JOBS - 3)                    # no room for another block's four products: launch what is there
    assert d.last_of_group(5)
    cfg.grad_ready = lambda params: None                        # a plain callback (no reducer): one group
    assert ops.DeferredWgrads(cfg, 11).policy == "all"
    monkeypatch.setenv("DEVIT_WGRAD_GROUP", "block")
    d = ops.DeferredWgrads(cfg, 11)
    assert all(d.last_of_group(i) for i in range(11))
    monkeypatch.setenv("DEVIT_WGRAD_GROUP", "bucket")           # asked for, but no reducer to align with
    assert ops.DeferredWgrads(cfg, 11).policy == "all"
    monkeypatch.setenv("DEVIT_WGRAD_GROUP", "nonsense")
    with pytest.raises(Exception, match="DEVIT_WGRAD_GROUP"):
        ops.DeferredWgrads(cfg, 11)
    # flush with nothing recorded reports the pending blocks and launches nothing (no GPU touched)
    monkeypatch.setenv("DEVIT_WGRAD_GROUP", "all")
    hook = Hook()
    cfg.grad_ready = hook
    d = ops.DeferredWgrads(cfg, 11)
    d.add(cfg.blocks[3], [])
    d.add(cfg.blocks[2], [])
    d.flush(512)
    assert hook.seen == [3, 2] and d.pending == []
