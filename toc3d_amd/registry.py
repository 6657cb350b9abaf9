"""Registry surface: ``ToC3DEVAViT`` / ``EVA_ViT`` on mmdet's BACKBONES, ``CPFPN`` on NECKS.

The reference registers its classes with ``@BACKBONES.register_module()`` (``toc3d_eva_vit.py:25``,
``eva_vit.py:270``) and ``@NECKS.register_module()`` (``cp_fpn.py:15``); configs name them by ``type=``.
When mmdet is importable the same names are registered there (``force=True`` so this package can shadow the
reference plugin); otherwise a shim registry with the same ``register_module()/build(cfg)`` API is used.
"""
from __future__ import annotations


class _ShimRegistry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **default_args):
        cfg = dict(cfg)
        cls = self.module_dict[cfg.pop("type")]
        cfg.update(default_args)
        return cls(**cfg)


try:  # pragma: no cover - mmdet is not installed in the build image
    from mmdet.models.builder import BACKBONES, NECKS
    HAVE_MMDET = True
except Exception:  # noqa: BLE001
    BACKBONES, NECKS = _ShimRegistry("backbone"), _ShimRegistry("neck")
    HAVE_MMDET = False


def register_all():
    from .backbone import EVA_ViT, ToC3DEVAViT
    from .neck import CPFPN
    for reg, cls in ((BACKBONES, ToC3DEVAViT), (BACKBONES, EVA_ViT), (NECKS, CPFPN)):
        try:
            reg.register_module(name=cls.__name__, force=True, module=cls)
        except TypeError:
            reg.register_module()(cls)


def build_backbone(cfg, **kw):
    return BACKBONES.build(cfg, **kw) if not HAVE_MMDET else BACKBONES.build(dict(cfg, **kw))


def build_neck(cfg, **kw):
    return NECKS.build(cfg, **kw) if not HAVE_MMDET else NECKS.build(dict(cfg, **kw))
