"""devit_amd: MI355X (gfx950) implementation of the DeViT hot path -- gated ViT block forward/backward and
the DEKD distillation step -- behind the reference's module / loss / CLI surface.

Importing the package needs neither a GPU nor the shared library; running any op needs both
(libdevit_hip.so, built by __graft_entry__.build(), and a gfx950 device) and fails loudly otherwise.
"""
from . import de_vit, losses, registry  # noqa: F401
from .de_vit import Attention, Block, Mlp, VisionTransformer, model_config  # noqa: F401
from .losses import (DistillLoss, LabelSmoothingCrossEntropy, SoftTargetCrossEntropy, feature_relation_loss,  # noqa: F401
                     relation_losses_packed)
from .registry import create_model, register_model  # noqa: F401

__version__ = "0.1.0"
