"""Model registry with timm 0.5.4's `register_model` / `create_model` semantics.

timm is not a dependency: the reference only uses its registry, PatchEmbed, DropPath and a few helpers on
this path.  If timm happens to be importable the models are registered there as well, so
`timm.models.create_model('dedeit', ...)` (distill_sub.py:212) resolves to this implementation.
"""
_REGISTRY = {}


def register_model(fn):
    _REGISTRY[fn.__name__] = fn          # later registrations override earlier ones, like timm
    try:                                 # pragma: no cover - timm is absent in the build image
        from timm.models.registry import register_model as _timm_register
        _timm_register(fn)
    except Exception:
        pass
    return fn


def is_model(name):
    return name in _REGISTRY


def list_models():
    return sorted(_REGISTRY)


def create_model(model_name, pretrained=False, checkpoint_path='', **kwargs):
    """timm.models.create_model: drops None-valued kwargs, unknown name -> RuntimeError."""
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    if model_name not in _REGISTRY:
        raise RuntimeError('Unknown model (%s)' % model_name)
    model = _REGISTRY[model_name](pretrained=pretrained, **kwargs)
    if checkpoint_path:
        import torch
        ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
        model.load_state_dict(ckpt['model'] if 'model' in ckpt else ckpt)
    return model
