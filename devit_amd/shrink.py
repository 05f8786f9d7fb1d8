"""Physical shrinking: turn the 0/1 (or real-valued) gates of a trained sub-model into smaller GEMMs.

The reference prunes by MASKING only: `core/imp_rank.py:50-71,132-153` writes `m.gate` for every `Mlp` / `Attention`
and `models/de_vit.py:42-43,77-79` multiplies the hidden activations / head outputs by it, so a sub-model shrunk by 30 %
still runs every FLOP of the dense model; the saving exists only in the analytic model of `core/compute_metric.py:44-59`
(9.20 -> 6.36 GFLOPs at 0.3/0.3).  `compact()` makes it real for inference:

  * Attention: heads with gate == 0 are dropped from the qkv GEMM (rows of `qkv.weight` / `qkv.bias`), from the attention
    grid and from the proj GEMM (columns of `proj.weight`); a non-zero gate g_h is folded into head h's proj columns.
  * Mlp: neurons with gate == 0 are dropped from fc1 (rows) and fc2 (columns); g_j is folded into fc2 column j.
  * Kernel granularity: the hidden width is padded to a multiple of 128 and the head count to an even number with
    all-zero units (zero weights and bias: gelu(0) = 0, and a head with q = k = v = 0 outputs 0), so the result is
    EXACTLY the masked model's function.

The fp32 master parameters, `state_dict()` and the gates are untouched; the compacted bf16 weights live beside them
(`block._compact`) and are used by `Block.block_params`.  Compacted blocks are inference-only (EncoderFn refuses to
record a backward through them); `uncompact()` drops them.  `neuron_output` / `head_output` of a compacted block hold
the kept units only.
"""
import torch

from . import ops

__all__ = ["compact", "uncompact", "compact_block_weights", "compacted_gflops", "masks_from_sparsity", "load_policy",
           "get_policy", "save_gates", "load_gates", "read_shrink_checkpoint", "rank_units", "apply_shrink", "neuron_scores", "head_scores"]


def _round_up(n, m):
    return (n + m - 1) // m * m


@torch.no_grad()
def compact_block_weights(blk, heads=True):
    """fp32 compacted weights of one Block from its current gates (pure tensor indexing: runs on any device).
    Returns dict(num_heads, qkv_w [3*64*Hr, D], qkv_b, proj_w [D, 64*Hr], fc1_w [Nr, D], fc1_b, fc2_w [D, Nr],
    kept_heads, kept_neurons).  heads=False: only the MLP is compacted; all heads stay in the qkv / proj GEMMs and the
    head gate keeps acting at run time (a block whose q/k/v of ALL heads are read, see compact())."""
    attn, mlp = blk.attn, blk.mlp
    dev = attn.qkv.weight.device
    D, H = attn.qkv.weight.shape[1], attn.num_heads
    hd = attn.qkv.weight.shape[0] // (3 * H)
    hg = attn.gate.detach().float().cpu().reshape(-1) if heads else torch.ones(H)
    keep_h = torch.nonzero(hg != 0).reshape(-1)
    Hr = max(2, _round_up(len(keep_h), 2))                       # heads run (3 * 64 * Hr must be a multiple of 128)
    qw = attn.qkv.weight.detach().float().view(3, H, hd, D)
    qb = attn.qkv.bias.detach().float().view(3, H, hd)
    pw = attn.proj.weight.detach().float().view(D, H, hd)
    kh = keep_h.to(dev)
    qkv_w = torch.zeros((3, Hr, hd, D), device=dev)
    qkv_b = torch.zeros((3, Hr, hd), device=dev)
    proj_w = torch.zeros((D, Hr, hd), device=dev)
    qkv_w[:, : len(keep_h)] = qw[:, kh]
    qkv_b[:, : len(keep_h)] = qb[:, kh]
    proj_w[:, : len(keep_h)] = pw[:, kh] * hg[keep_h].to(dev)[None, :, None]
    ng = mlp.gate.detach().float().cpu().reshape(-1)
    keep_n = torch.nonzero(ng != 0).reshape(-1)
    Nr = max(128, _round_up(len(keep_n), 128))
    kn = keep_n.to(dev)
    fc1_w = torch.zeros((Nr, D), device=dev)
    fc1_b = torch.zeros((Nr,), device=dev)
    fc2_w = torch.zeros((mlp.fc2.weight.shape[0], Nr), device=dev)
    fc1_w[: len(keep_n)] = mlp.fc1.weight.detach().float()[kn]
    fc1_b[: len(keep_n)] = mlp.fc1.bias.detach().float()[kn]
    fc2_w[:, : len(keep_n)] = mlp.fc2.weight.detach().float()[:, kn] * ng[keep_n].to(dev)[None, :]
    return dict(num_heads=Hr, qkv_w=qkv_w.reshape(3 * Hr * hd, D).contiguous(), qkv_b=qkv_b.reshape(-1).contiguous(),
                proj_w=proj_w.reshape(D, Hr * hd).contiguous(), fc1_w=fc1_w, fc1_b=fc1_b, fc2_w=fc2_w,
                kept_heads=keep_h.tolist(), kept_neurons=keep_n)


_serial = [0]          # stamps every compaction (cache keys use it, never id())


@torch.no_grad()
def compact(model, trainable=False):
    """Build the compacted weights of every block of `model` (a devit_amd VisionTransformer, MultiViT sub-models
    included) from its current gates.  Returns [(kept heads, heads run, kept neurons, neurons run)] per block.

    trainable=True: the model can be TRAINED through the compacted blocks -- what distill_sub.py:384-401 does with the
    masked model (rank one batch, set the gates, train), at the shrunk model's FLOPs.  Needs 0/1 gates (the masks of
    core/imp_rank.py:50-71,132-153).  The fp32 masters stay the parameters: every training forward re-gathers the kept
    rows / columns from their current bf16 copies (refresh_compact), the backward's weight-gradient GEMMs run at the
    compact shapes and their results are added into the kept rows / columns of the masters' gradients; masked units get
    exact-zero gradients, as in the masked model (their activations are multiplied by a zero gate there)."""
    report = []
    _serial[0] += 1
    # DEKD reads q/k/v of ALL heads of the middle block (engine.py:91-92; the gate of models/de_vit.py:77-79 acts on the head
    # OUTPUTS, so a masked head's q/k/v still enter -- and get gradients from -- the relation loss): when training, that
    # block keeps every head in its GEMMs and masks at run time
    keep_heads = set()
    if trainable:
        for m in model.modules():
            if type(m).__name__ == "VisionTransformer" and len(m.blocks) >= 2:
                keep_heads.add(id(m.blocks[len(m.blocks) // 2 - 1]))
    for blk in _blocks(model):
        if trainable:
            for g in (blk.attn.gate, blk.mlp.gate):
                if not bool(((g == 0) | (g == 1)).all()):
                    raise ValueError("shrink.compact(trainable=True) needs 0/1 gates (imp_rank masks); real-valued gates are "
                                     "folded into the weights for inference only")
        heads = id(blk) not in keep_heads
        w = compact_block_weights(blk, heads=heads)
        dev = w["qkv_w"].device
        blk._compact_version = getattr(blk, "_compact_version", 0) + 1      # Block.block_params' cache key (never id())
        blk._compact = dict(num_heads=w["num_heads"], qkv_b=w["qkv_b"], fc1_b=w["fc1_b"],
                            qkv_w16=ops.cast_bf16(w["qkv_w"], None), proj_w16=ops.cast_bf16(w["proj_w"], None),
                            fc1_w16=ops.cast_bf16(w["fc1_w"], None), fc2_w16=ops.cast_bf16(w["fc2_w"], None),
                            kept_heads=w["kept_heads"], kept_neurons=w["kept_neurons"], trainable=bool(trainable),
                            serial=_serial[0] * 100 + len(report),
                            heads_compacted=heads,
                            neurons_idx=w["kept_neurons"].to(dev))
        # the dense k-major copy of fc2's weight (de_vit._w16t, made by an earlier dense forward) is never read while the block runs compacted:
        # dropped, so that FlatParams.refresh_kmajor() stops re-transposing it after every optimizer step (advisor r05); a dense forward after
        # uncompact() makes it again
        blk.mlp.fc2.__dict__.pop("_w16t", None)
        c = blk._compact
        if c["fc2_w16"].is_cuda and c["fc2_w16"].shape[0] == 384:      # fc2's forward on the full-row GEMM (csrc/gemm.hip): k-major copy
            c["fc2_w16t"] = torch.empty((c["fc2_w16"].shape[1], 384), dtype=c["fc2_w16"].dtype, device=dev)
            ops.transpose16(c["fc2_w16"], c["fc2_w16t"])
        report.append((len(w["kept_heads"]), w["num_heads"], len(w["kept_neurons"]), w["fc1_w"].shape[0]))
    return report


class _JobTable:
    """A device-resident table of devit_index_job entries (include/devit_hip.h) + what keeps its pointers alive."""

    def __init__(self, jobs, keep, device):
        import ctypes as C
        from . import _lib as L
        arr = (L.IndexJob * len(jobs))(*jobs)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table = host.to(device)
        self.n, self.keep = len(jobs), keep
        self.biggest = max(j.rows * j.cols for j in jobs)

    def run(self):
        from ._lib import call, ptr, stream_ptr
        call("devit_index_copy", ptr(self.table), self.n, max(1, min(64, (self.biggest + 4095) // 4096)), stream_ptr())


def _job(src, dst, idx, rows, cols, src_ld, dst_ld, mode, elem):
    from . import _lib as L
    return L.IndexJob(src.data_ptr(), dst.data_ptr(), idx.data_ptr(), rows, cols, src_ld, dst_ld, mode, elem)


def _index_maps(blk, c):
    """int32 device maps compact unit -> master unit (-1: padding) for the block's four shapes of copy: qkv rows
    [3 * Hr * 64], proj columns [Hr * 64], fc1 rows / fc2 columns [Nr]."""
    m = c.get("maps")
    if m is None:
        dev = c["qkv_w16"].device
        H, Hr, hd = blk.attn.num_heads, c["num_heads"], 64
        heads = list(c["kept_heads"])
        hmap = torch.full((Hr,), -1, dtype=torch.int64)
        hmap[: len(heads)] = torch.as_tensor(heads, dtype=torch.int64)
        e = torch.arange(hd)
        cols = torch.where(hmap[:, None] >= 0, hmap[:, None] * hd + e[None, :], torch.full((Hr, hd), -1)).reshape(-1)
        rows = torch.cat([torch.where(cols >= 0, cols + j * H * hd, cols) for j in range(3)])
        Nr = c["fc1_w16"].shape[0]
        nmap = torch.full((Nr,), -1, dtype=torch.int64)
        nmap[: c["neurons_idx"].numel()] = c["neurons_idx"].cpu()
        m = c["maps"] = dict(qkv_rows=rows.to(torch.int32).to(dev), head_cols=cols.to(torch.int32).to(dev),
                             neurons=nmap.to(torch.int32).to(dev))
    return m


@torch.no_grad()
def refresh_blocks(blocks):
    """Re-gather the 16-bit weights and fp32 biases of every trainable compacted block in `blocks` from the masters' current
    values (the fused optimizer rewrites the masters and their bf16 copies in place every step): ONE launch over a job
    table (devit_index_copy); the table is rebuilt when a source or destination moved.  Gates are 0/1: nothing to fold.
    Blocks in eval mode are refreshed once after training and whenever torch rewrote a master since the last gather (load_state_dict, an EMA
    swap, broadcast_module: the parameters' version counters), otherwise left alone."""
    from .de_vit import _w16

    def versions(blk):
        # torch-side rewrites of the masters (load_state_dict / --resume, an EMA swap, ddp.broadcast_module) bump these; the
        # fused optimizer writes through raw pointers and does not -- that case is `blk.training` below
        # (`p.data = other` -- how EMA weights are usually swapped in, and what module._apply / .to() does -- may leave the counter alone:
        # the storage address is part of the key; proj.bias / fc2.bias are read in place, never gathered)
        return tuple((p._version, p.data_ptr()) for p in (blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight, blk.mlp.fc1.weight,
                                                          blk.mlp.fc1.bias, blk.mlp.fc2.weight))
    todo = []
    for blk in blocks:
        c = getattr(blk, "_compact", None)
        if c is None or not c.get("trainable"):
            continue
        ver = versions(blk)
        if blk.training or c.get("stale_after_training") or c.get("master_versions") != ver:
            c["stale_after_training"] = blk.training       # the first eval forward after training re-gathers once more
            c["master_versions"] = ver
            todo.append((blk, c))
    if not todo:
        return
    srcs = [(_w16(b.attn.qkv), _w16(b.attn.proj), _w16(b.mlp.fc1), _w16(b.mlp.fc2), b.attn.qkv.bias, b.mlp.fc1.bias) for b, _ in todo]
    key = tuple(t.data_ptr() for ss in srcs for t in ss) + tuple(c["serial"] for _, c in todo)
    owner = todo[0][1]
    tab = owner.get("refresh_table")
    if tab is None or tab[0] != key:
        jobs, keep = [], []
        for (blk, c), (q16, p16, f1, f2, qb, f1b) in zip(todo, srcs):
            m = _index_maps(blk, c)
            D, Hr = blk.attn.qkv.weight.shape[1], c["num_heads"]
            Nr, Hid = c["fc1_w16"].shape[0], blk.mlp.fc1.weight.shape[0]
            jobs += [_job(q16, c["qkv_w16"], m["qkv_rows"], 3 * Hr * 64, D, D, D, 0, 2),
                     _job(p16, c["proj_w16"], m["head_cols"], D, Hr * 64, p16.shape[1], Hr * 64, 1, 2),
                     _job(f1, c["fc1_w16"], m["neurons"], Nr, D, D, D, 0, 2),
                     _job(f2, c["fc2_w16"], m["neurons"], f2.shape[0], Nr, Hid, Nr, 1, 2),
                     _job(qb, c["qkv_b"], m["qkv_rows"], 3 * Hr * 64, 1, 1, 1, 0, 4),
                     _job(f1b, c["fc1_b"], m["neurons"], Nr, 1, 1, 1, 0, 4)]
            keep.append((q16, p16, f1, f2, qb, f1b, m))
        # (second launch, behind the gathers: the k-major copies of the compact fc2 weights the full-row GEMM reads, ops._Transposes)
        tr = ops._Transposes([(c["fc2_w16"], c["fc2_w16t"]) for _, c in todo if c.get("fc2_w16t") is not None]) \
            if any(c.get("fc2_w16t") is not None for _, c in todo) else None
        tab = owner["refresh_table"] = (key, _JobTable(jobs, keep, owner["qkv_w16"].device), tr)
    tab[1].run()
    if tab[2] is not None:
        tab[2].run()


def attach_training(blk, bp, c):
    """Make a compacted BlockParams trainable: compact-shaped gradient slots where the GEMM weights stand, the masters as the
    block's parameters, and the scatter of the slots into the masters' gradients after the block's backward."""
    attn, mlp = blk.attn, blk.mlp
    dev = c["qkv_w16"].device
    D, H, Hr, hd = attn.qkv.weight.shape[1], attn.num_heads, c["num_heads"], 64
    Nr, Hid = c["fc1_w16"].shape[0], mlp.fc1.weight.shape[0]
    slots = c.get("slots")
    if slots is None:
        slots = c["slots"] = dict(qkv_w=ops.GradSlot((3 * Hr * hd, D), dev), qkv_b=ops.GradSlot((3 * Hr * hd,), dev),
                                  proj_w=ops.GradSlot((D, Hr * hd), dev), fc1_w=ops.GradSlot((Nr, D), dev),
                                  fc1_b=ops.GradSlot((Nr,), dev), fc2_w=ops.GradSlot((D, Nr), dev))
    bp.qkv_w, bp.proj_w, bp.fc1_w, bp.fc2_w = slots["qkv_w"], slots["proj_w"], slots["fc1_w"], slots["fc2_w"]
    # the compact biases are plain fp32 tensors read by the forward; their gradients go to slots
    bp.qkv_b, bp.fc1_b = _BiasWithSlot(c["qkv_b"], slots["qkv_b"]), _BiasWithSlot(c["fc1_b"], slots["fc1_b"])
    bp.masters = [blk.norm1.weight, blk.norm1.bias, attn.qkv.weight, attn.qkv.bias, attn.proj.weight, attn.proj.bias,
                  blk.norm2.weight, blk.norm2.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias]
    m = _index_maps(blk, c)

    def finish():
        """Compact weight gradients -> the kept rows / columns of the masters' gradients (one launch), slots zeroed."""
        if slots["qkv_w"].grad is None:
            return
        g = ops.grad_buf
        with torch.no_grad():
            dsts = [g(attn.qkv.weight), g(attn.proj.weight), g(mlp.fc1.weight), g(mlp.fc2.weight), g(attn.qkv.bias), g(mlp.fc1.bias)]
            srcs = [g(slots[k]) for k in ("qkv_w", "proj_w", "fc1_w", "fc2_w", "qkv_b", "fc1_b")]
            key = tuple(t.data_ptr() for t in dsts + srcs)
            tab = c.get("scatter_table")
            if tab is None or tab[0] != key:
                jobs = [_job(srcs[0], dsts[0], m["qkv_rows"], 3 * Hr * hd, D, D, D, 2, 4),
                        _job(srcs[1], dsts[1], m["head_cols"], D, Hr * hd, Hr * hd, H * hd, 3, 4),
                        _job(srcs[2], dsts[2], m["neurons"], Nr, D, D, D, 2, 4),
                        _job(srcs[3], dsts[3], m["neurons"], D, Nr, Nr, Hid, 3, 4),
                        _job(srcs[4], dsts[4], m["qkv_rows"], 3 * Hr * hd, 1, 1, 1, 2, 4),
                        _job(srcs[5], dsts[5], m["neurons"], Nr, 1, 1, 1, 2, 4)]
                tab = c["scatter_table"] = (key, _JobTable(jobs, (dsts, srcs, m), dev))
            tab[1].run()                 # (the add modes zero the slots they consumed)
    bp.finish = finish


class _BiasWithSlot:
    """A compact fp32 bias as the kernels read it (data_ptr) whose gradient accumulates in a GradSlot."""
    __slots__ = ("value", "slot")

    def __init__(self, value, slot):
        self.value, self.slot = value, slot

    def data_ptr(self):
        return self.value.data_ptr()

    shape = property(lambda self: self.value.shape)
    device = property(lambda self: self.value.device)
    requires_grad = False

    @property
    def grad(self):
        return self.slot.grad

    @grad.setter
    def grad(self, v):
        self.slot.grad = v


def uncompact(model):
    for blk in _blocks(model):
        blk._compact = None
        blk._compact_version = getattr(blk, "_compact_version", 0) + 1


def _blocks(model):
    from .de_vit import Block
    return [m for m in model.modules() if isinstance(m, Block)]


def compacted_gflops(model, tokens=198, patch_dim=768, num_classes=None):
    """Forward GFLOPs per image (2 FLOP per MAC, the accounting of BASELINE.md §2) of the model as it would run:
    compacted blocks at their run sizes, the others dense."""
    total = 0.0
    for blk in _blocks(model):
        D = blk.attn.qkv.weight.shape[1]
        c = getattr(blk, "_compact", None)
        Da = c["qkv_w16"].shape[0] // 3 if c else D
        Hd = c["fc1_w16"].shape[0] if c else blk.mlp.fc1.weight.shape[0]
        total += 2.0 * tokens * (D * 3 * Da + Da * D + 2 * D * Hd) + 4.0 * tokens * tokens * Da
    for m in model.modules():
        if type(m).__name__ == "VisionTransformer":
            D = m.embed_dim
            total += 2.0 * (tokens - (2 if getattr(m, "dist_token", None) is not None else 1)) * patch_dim * D
            nc = num_classes if num_classes is not None else getattr(m, "num_classes", 0)
            total += 2.0 * D * nc * (2 if getattr(m, "head_dist", None) is not None else 1)
    return total / 1e9


def masks_from_sparsity(model, neuron_sparsity, head_sparsity, neuron_rank, head_rank):
    """core/imp_rank.py:50-62 (`mlp_neuron_mask`) and :132-144 (`attn_head_mask`): per block keep the
    int(n * (1 - ratio)) highest-ranked units; `*_rank[i]` is an ascending argsort of importance scores."""
    policy = []
    for i, blk in enumerate(_blocks(model)):
        nh, nn_ = blk.attn.num_heads, blk.mlp.hidden_features
        hm, nm = torch.zeros(nh), torch.zeros(nn_)
        keep_h = int(nh * (1 - head_sparsity[i]))
        keep_n = int(nn_ * (1 - neuron_sparsity[i]))
        hm[list(head_rank[i])[::-1][:keep_h]] = 1
        nm[list(neuron_rank[i])[::-1][:keep_n]] = 1
        policy.append((hm, nm))
    return policy


def load_policy(model, policy):
    """Assign gates from a shrink policy: a sequence over blocks of (head_mask [H], neuron_mask [hidden]) arrays -- the
    content of the reference's `shrinked_policy.npy` (core/imp_rank.py writes one 0/1 vector per Attention / Mlp in
    module order)."""
    blocks = _blocks(model)
    if len(policy) != len(blocks):
        raise ValueError(f"policy has {len(policy)} entries for {len(blocks)} blocks")
    for blk, (hm, nm) in zip(blocks, policy):
        blk.attn.gate = torch.as_tensor(hm, dtype=torch.float32).reshape(-1).clone()
        blk.mlp.gate = torch.as_tensor(nm, dtype=torch.float32).reshape(-1).clone()


def get_policy(model):
    """[(head gate [H], neuron gate [hidden])] per block, CPU float tensors."""
    return [(b.attn.gate.detach().float().cpu().clone(), b.mlp.gate.detach().float().cpu().clone()) for b in _blocks(model)]


def save_gates(model, path):
    """Persist the gates beside a checkpoint.  The reference keeps `gate` out of `state_dict()` (plain attributes,
    models/de_vit.py:33,63), so a shrunk sub-model reloaded from `checkpoint.pth` runs dense again;
    the state_dict ABI stays as it is and the gates travel in their own file."""
    torch.save([(h.numpy(), n.numpy()) for h, n in get_policy(model)], path)


def load_gates(model, path):
    load_policy(model, [(torch.from_numpy(h), torch.from_numpy(n)) for h, n in torch.load(path, weights_only=False)])


# ------------------------------------------------------------------------------------------------------
# the training CLI's shrink step (distill_sub.py:383-401): policy files -> one-batch importance ranking -> gates
# ------------------------------------------------------------------------------------------------------
def read_shrink_checkpoint(path):
    """distill_sub.py:384-389: `shrinked_policy.npy` [population, 24 or 25] and `shrinked_accuracy.npy` [population]
    written by the reference's shrink.py:417-418; the best-accuracy row gives 12 neuron ratios (columns 0..11) and the
    head ratios.  The reference slices the heads as `[12:-1]` -- 11 values from a 24-column file, so its attn_head_mask
    would run out at the last block: columns 12..23 are used here whether the file has 24 or 25."""
    import os
    import numpy as np
    policy = np.load(os.path.join(path, 'shrinked_policy.npy'))
    acc = np.load(os.path.join(path, 'shrinked_accuracy.npy'))
    if policy.ndim != 2 or policy.shape[1] not in (24, 25) or policy.shape[0] != acc.reshape(-1).shape[0]:
        raise ValueError(f"shrink checkpoint {path}: policy {policy.shape} / accuracy {acc.shape} are not [pop, 24|25] / [pop]")
    row = policy[int(np.argmax(acc))]
    return row[:12].astype(float), row[12:24].astype(float)


def _center(K):
    return K - K.mean(-2, keepdim=True) - K.mean(-1, keepdim=True) + K.mean((-2, -1), keepdim=True)


def _gauss_mix(x):
    """core/imp_rank.py:182-192,229: mean of five Gaussian kernels (sigma 1, 2, 4, 8, 16) on the rows of x [..., B, F]."""
    inner = x @ x.transpose(-1, -2)
    nrm = torch.diagonal(inner, dim1=-2, dim2=-1)
    d2 = nrm.unsqueeze(-1) + nrm.unsqueeze(-2) - 2 * inner
    return sum(torch.exp(-d2 / (2.0 * s * s)) for s in (1.0, 2.0, 4.0, 8.0, 16.0)) / 5.0


def _hsic(x, y, y_kernel, mean_sub):
    """core/imp_rank.py:203-239 (HSICLoss.forward), batched over leading dims of x.  x [..., B, F], y [B, C] or
    [..., B, F'].  `x - mean / (std + 1e-12)` keeps the reference's operator precedence."""
    if mean_sub:
        x = x - x.mean(-2, keepdim=True) / (x.std(-2, keepdim=True) + 1e-12)
        y = y - y.mean(-2, keepdim=True)
    gx = _center(_gauss_mix(x))
    gy = _center(y @ y.transpose(-1, -2)) if y_kernel == 'linear' else _center(_gauss_mix(y))
    return (gx * gy.transpose(-1, -2)).sum((-2, -1))         # trace(G_X G_Y)


@torch.no_grad()
def neuron_scores(neuron_output, prob):
    """core/imp_rank.py:31-41 for one Mlp: 0.1 * min-max(HSIC(activation of neuron j over [B, N], softmax(logits))) +
    0.9 * min-max(sum |activation|).  neuron_output [B, N, hidden], prob [B, C] -> [hidden]."""
    X = neuron_output.float()
    hs = _hsic(X.permute(2, 0, 1), prob, 'linear', True)
    hs = (hs - hs.min()) / (hs.max() - hs.min())
    act = X.abs().sum((0, 1))
    act = (act - act.min()) / (act.max() - act.min())
    return 0.1 * hs + 0.9 * act


@torch.no_grad()
def head_scores(head_output, prob):
    """core/imp_rank.py:108-123 for one Attention: relevance(head) - 0.1 * mean redundancy against the other heads, on
    the head's channel mean.  head_output [B, N, H, hd], prob [B, C] -> [H]."""
    Hh = head_output.float().mean(-1).permute(2, 0, 1)              # [H, B, N]
    relv = _hsic(Hh, prob, 'linear', True)
    nH = Hh.shape[0]
    red = torch.stack([sum(_hsic(Hh[a], Hh[b], 'rbf', False) for b in range(nH) if b != a) / (nH - 1) for a in range(nH)])
    return relv - 0.1 * red


@torch.no_grad()
def rank_units(model, data_loader, device=None):
    """One-batch importance ranking of every block's MLP neurons and attention heads (core/imp_rank.py:16-47 and
    :93-129).  Reads the `neuron_output` / `head_output` the forward leaves on the modules (post-gate values, SURVEY
    App. D Q3).  Returns (neuron_rank, head_rank): per block an ascending argsort (numpy), the input of
    masks_from_sparsity.  The reference calls the model in whatever mode it is in (train at that point of
    distill_sub.py, where a distilled model returns a tuple that F.softmax cannot take); here: eval mode, averaged heads.
    Host-side torch arithmetic on a single batch: a one-off setup step, not the hot path."""
    import numpy as np
    data, _ = next(iter(data_loader))
    if device is not None:
        data = data.to(device)
    was_training = model.training
    model.eval()
    out = model(data)
    out = (out[0] + out[1]) / 2 if isinstance(out, tuple) else out
    prob = torch.softmax(out.float(), dim=-1)
    neuron_rank, head_rank = [], []
    for blk in _blocks(model):
        neuron_rank.append(np.argsort(neuron_scores(blk.mlp.neuron_output, prob).cpu().numpy()))
        head_rank.append(np.argsort(head_scores(blk.attn.head_output, prob).cpu().numpy()))
    model.train(was_training)
    return neuron_rank, head_rank


def apply_shrink(model, data_loader, shrink_checkpoint, neuron_shrinking, head_shrinking, device=None):
    """distill_sub.py:383-401.  Returns the (head_mask, neuron_mask) policy it set.  A flag that cannot be honoured
    raises -- the reference dies on an undefined `neuron_sparsity` when --neuron_shrinking comes without
    --shrink_checkpoint; it never trains dense silently."""
    if not (neuron_shrinking or head_shrinking):
        return None
    if not shrink_checkpoint:
        raise ValueError("--neuron_shrinking / --head_shrinking need --shrink_checkpoint DIR holding shrinked_policy.npy and "
                         "shrinked_accuracy.npy (the reference reads the sparsity ratios from there, distill_sub.py:384-389)")
    neuron_sparsity, head_sparsity = read_shrink_checkpoint(shrink_checkpoint)
    neuron_rank, head_rank = rank_units(model, data_loader, device)
    blocks = _blocks(model)
    zeros = [0.0] * len(blocks)
    policy = masks_from_sparsity(model, neuron_sparsity if neuron_shrinking else zeros,
                                 head_sparsity if head_shrinking else zeros, neuron_rank, head_rank)
    load_policy(model, policy)
    return policy
