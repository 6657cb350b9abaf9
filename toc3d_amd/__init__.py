"""toc3d_amd -- MI355X (gfx950) implementation of the ToC3D / EVA-02 ViT backbone hot path.

Importing the package registers ``ToC3DEVAViT``, ``EVA_ViT`` (BACKBONES) and ``CPFPN`` (NECKS) under the
reference's type names, like ``projects/mmdet3d_plugin`` does on import (``tools/test.py:133-145``).
"""
from .backbone import EVA_ViT, ToC3DEVAViT, ToC3DViTReturnType
from .neck import CPFPN
from .preprocess import prepare_images
from .memory import TemporalMemory
from .head_tokens import HeadTokenEmbedding
from .registry import BACKBONES, NECKS, build_backbone, build_neck, register_all

register_all()

__all__ = ["ToC3DEVAViT", "EVA_ViT", "CPFPN", "ToC3DViTReturnType", "BACKBONES", "NECKS", "build_backbone", "build_neck", "prepare_images", "TemporalMemory", "HeadTokenEmbedding"]
