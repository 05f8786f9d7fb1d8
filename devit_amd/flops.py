"""Analytic FLOP / parameter accounting of the gated ViT (2 FLOP per MAC), restated from the reference's own formulas
(core/compute_metric.py:1-28 `cal_shrink_paras`, :31-64 `cal_shrink_flops`, :67-69 `cal_shrink_macs`) and pinned to their
values by tests/golden/flops.json: `dedeit` 9.197764608 GFLOPs / 22.03684 M parameters (N = 197, 1000 classes; the 9.19 of
core/shrink_imp.py:144), DeiT-B 35.127656448 / 86.540008, `dedeit` at 0.3 / 0.3 sparsity 6.355820544.

bench.py's roofline uses the same formula at the benchmark's geometry (N = 198 tokens of the distilled models, C = 25):
teacher forward 35.311, student forward 9.247 GFLOP per image."""

__all__ = ["forward_gflops", "params_m", "macs_g", "step_gflops_per_image", "RELATION_LOSS_GFLOP", "lean_tail_skipped_gflops",
           "step_gflops_per_image_executed"]

# q/k/v feature-relation losses per image: Gram matrices of both models forward (3 components x
# 2 * 198^2 * (768 + 384)) and the student-side backward
RELATION_LOSS_GFLOP = 0.271 + 0.181


def _per_layer(neuron_sparsity, head_sparsity, layer):
    ns = list(neuron_sparsity) if neuron_sparsity is not None else [0.0] * layer
    hs = list(head_sparsity) if head_sparsity is not None else [0.0] * layer
    assert len(hs) == layer and len(ns) == layer, 'The number of layer is not equal to the number of head sparsity.'
    return ns, hs


def forward_gflops(emb=768, seq_length=197, mlp_ratio=4, head=12, layer=12, num_class=1000, neuron_sparsity=None,
                   head_sparsity=None):
    """core/compute_metric.py:31-64: patch embedding 2 * 3 * emb * 224^2, per block the kept heads' qkv projection,
    q k^T, (q k^T) v and output projection plus the kept neurons' two MLP matrices, one classifier head; softmax and
    norms neglected.  Kept heads = int((1 - s) * head), kept neurons = int(mlp_ratio * (1 - s) * emb)."""
    ns, hs = _per_layer(neuron_sparsity, head_sparsity, layer)
    head_dim = emb / head
    flops = 2 * 3 * emb * 224 ** 2
    for n_s, h_s in zip(ns, hs):
        sa = 3 * 2 * seq_length * emb * head_dim + 2 * head_dim * seq_length ** 2 + 2 * head_dim * seq_length ** 2
        kept_heads = int((1 - h_s) * head)
        hidden = int(mlp_ratio * (1 - n_s) * emb)
        flops += sa * kept_heads + seq_length * 2 * head_dim * kept_heads * emb
        flops += seq_length * hidden * 2 * emb + seq_length * emb * 2 * hidden
    flops += 2 * emb * num_class
    return flops / 1e9


def macs_g(**kw):
    """core/compute_metric.py:67-69."""
    return forward_gflops(**kw) / 2


def params_m(emb=768, seq_length=197, mlp_ratio=4, head=12, layer=12, num_class=1000, neuron_sparsity=None,
             head_sparsity=None):
    """core/compute_metric.py:1-28 (millions of parameters; one class token, one head, as the reference counts)."""
    ns, hs = _per_layer(neuron_sparsity, head_sparsity, layer)
    head_dim = emb / head
    paras = emb * 3 * 16 ** 2 + emb + seq_length * emb + emb
    ln = 2 * emb
    for n_s, h_s in zip(ns, hs):
        kept_heads = int((1 - h_s) * head)
        hidden = int(mlp_ratio * (1 - n_s) * emb)
        mhsa = kept_heads * 3 * emb * head_dim + kept_heads * head_dim * emb + emb
        mlp = 2 * emb * hidden + emb + hidden
        paras += ln + mhsa + ln + mlp
    paras += ln + emb * num_class + num_class
    return paras / 1e6


def step_gflops_per_image(num_class=25, tokens=198):
    """One distill_sub step per image: DeiT-B teacher forward + `dedeit` student forward and backward (backward = 2 x the
    forward GEMM FLOPs) + the relation losses: 35.311 + 3 * 9.247 + 0.452 = 63.503 at C = 25."""
    teacher = forward_gflops(seq_length=tokens, num_class=num_class)
    student = forward_gflops(emb=384, head=6, seq_length=tokens, num_class=num_class)
    return teacher + 3.0 * student + RELATION_LOSS_GFLOP


def lean_tail_skipped_gflops(emb=768, mlp_ratio=4, tokens=198, ntok=2):
    """Forward GFLOPs per image of the LAST block that the lean tail (devit_amd.de_vit.lean_tail) does not execute: the
    Q projection, the attention of, the output projection and both MLP matrices on the tokens - ntok rows whose results
    nothing reads (models/de_vit.py:286-288 keeps x[:, 0] and x[:, 1]); K and V of every row are still computed."""
    dead = tokens - ntok
    hidden = int(mlp_ratio * emb)
    return (2.0 * dead * emb * emb            # Q projection
            + 4.0 * dead * tokens * emb       # q k^T and (q k^T) v over all heads
            + 2.0 * dead * emb * emb          # output projection
            + 4.0 * dead * emb * hidden) / 1e9


def step_gflops_per_image_executed(num_class=25, tokens=198):
    """step_gflops_per_image minus what the lean last blocks skip (teacher forward; student forward and backward):
    the FLOPs the step EXECUTES.  63.503 - 2.431 - 3 * 0.638 = 59.159 at C = 25."""
    return (step_gflops_per_image(num_class, tokens) - lean_tail_skipped_gflops(768, 4, tokens, 2)
            - 3.0 * lean_tail_skipped_gflops(384, 4, tokens, 2))
