"""Optimizer tail of the DEKD step on flat buffers: global-norm clip + AdamW + EMA + bf16 weight re-cast in two
launches (devit_sumsq_f32, devit_adamw_step).  Replaces timm NativeScaler -> clip_grad_norm_ -> AdamW.step and
ModelEma.update (engine.py:127,131-132; ~600 tiny kernels in the reference)."""
import torch

from . import _lib as L
from ._lib import call, ptr, stream_ptr


def no_decay_names(model):
    """Parameters timm's create_optimizer (distill_sub.py:340) puts into its weight_decay = 0 group whenever
    weight_decay > 0: 1-D tensors, `.bias`, and the names in model.no_weight_decay()."""
    skip = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()
    return {n for n, p in model.named_parameters()
            if p.requires_grad and (p.ndim <= 1 or n.endswith(".bias") or n in skip)}


class FlatAdamW:
    """AdamW over ddp.FlatParams.  `no_decay`: parameter names exempt from weight decay (see no_decay_names; every
    tensor starts on a 4-element granule of the flat buffer, so the exemption is one byte per granule)."""

    def __init__(self, flat, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=None, ema_decay=None,
                 no_decay=None):
        self.flat, self.betas, self.eps, self.weight_decay = flat, betas, eps, weight_decay
        self.max_norm, self.ema_decay = max_norm, ema_decay
        dev = flat.flat.device
        self.no_decay4 = None
        if weight_decay and no_decay:
            mask = torch.zeros(flat.numel // 4, dtype=torch.uint8)
            for name, p, o in zip(flat.names, flat.params, flat.offsets):
                if name in no_decay:
                    mask[o // 4:(o + p.numel() + 3) // 4] = 1
            self.no_decay4 = mask.to(dev)
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.ema = flat.flat.clone() if ema_decay is not None else None
        self.step_count = 0
        self.param_groups = [{"lr": lr, "params": flat.params}]
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        # per-step scalars (lr, bias corrections) travel through pinned memory with an async copy; the host runs ahead of
        # the GPU, so every step in flight needs its own slot (a ring far deeper than the launch queue)
        self._dyn_host = torch.zeros((256, 3), dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros((256, 3))
        self._dyn = torch.zeros(3, dtype=torch.float32, device=dev)
        self._ws = torch.empty(4096, dtype=torch.uint8, device=dev)

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self):
        L.require_device(self.flat.flat)
        self.step_count += 1
        b1, b2 = self.betas
        slot = self._dyn_host[self.step_count % self._dyn_host.shape[0]]
        slot[0] = self.param_groups[0]["lr"]
        slot[1] = 1.0 - b1 ** self.step_count
        slot[2] = 1.0 - b2 ** self.step_count
        self._dyn.copy_(slot, non_blocking=True)
        f = self.flat
        clip = self.max_norm is not None and self.max_norm > 0
        if clip:
            call("devit_sumsq_f32", ptr(f.flat_grad), f.numel, ptr(self.gnorm_sq), ptr(self._ws), self._ws.numel(),
                 stream_ptr())
        # f.grad_scale: 1 / world after the summing bucket all-reduce (ddp.BucketedGradReducer.finish) -- the mean of DDP
        call("devit_adamw_step", ptr(f.flat), ptr(f.flat_grad), ptr(self.m), ptr(self.v), ptr(self.ema), ptr(f.flat16),
             ptr(self.no_decay4), ptr(self.gnorm_sq) if clip else None, ptr(self._dyn), f.numel, b1, b2, self.eps,
             self.weight_decay, float(self.max_norm or 0.0), float(self.ema_decay or 0.0), float(f.grad_scale),
             stream_ptr())
        f.grad_scale = 1.0
        f.refresh_kmajor()      # the kernel rewrote flat16 in place: the k-major weight copies derived from it follow (one launch)

    def ema_state_dict(self, model):
        """EMA weights keyed like model.state_dict() (timm get_state_dict(model_ema), distill_sub.py:429)."""
        if self.ema is None:
            return None
        f, out = self.flat, {}
        by_id = {id(p): n for n, p in model.named_parameters()}
        for p, o in zip(f.params, f.offsets):
            out[by_id[id(p)]] = self.ema[o:o + p.numel()].view_as(p).clone()
        return {k: out[k] for k in model.state_dict() if k in out}

    def state_dict(self):
        return {"m": self.m, "v": self.v, "ema": self.ema, "step": self.step_count, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        if self.ema is not None and sd.get("ema") is not None:
            self.ema.copy_(sd["ema"])
        self.step_count = sd["step"]
        self.param_groups[0]["lr"] = sd["lr"]
