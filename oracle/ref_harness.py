"""TEST INFRASTRUCTURE -- import the real reference backbone on CPU (build container only).

The reference (``/root/reference``, read-only) is pure Python but depends on
mmcv / mmdet / detectron2 / timm / fairscale, none of which are installed.
For the hot path only their *registries and trivial helpers* are touched, so
this harness installs minimal stand-in modules in ``sys.modules`` and imports
the backbone files where they lie.  Nothing from the
reference is copied into the repo, and this module refuses to work when
``/root/reference`` is absent (e.g. on the GPU box).

Harness patches (outside the reference's files), needed to make the
reference a deterministic function:

* ``torch.sort`` -> ``stable=True``       (tie rule: lowest index first;
  reference: ``toc3d_utils.py:139`` sorts without ``stable``),
* ``F.gumbel_softmax`` -> ``softmax(logits + g)`` with ``g`` injected per call
  (reference: ``toc3d_utils.py:147`` draws noise even in eval).
"""
from __future__ import annotations

import contextlib
import importlib
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get("TOC3D_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF, "projects", "mmdet3d_plugin"))


class _Registry:
    """Stands in for an mmcv Registry: ``@X.register_module()`` just records the class."""

    def __init__(self):
        self.module_dict = {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.module_dict[cls.__name__] = cls
            return cls
        return deco


_LOADED = {}


def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def load_reference():
    """Returns a namespace with the reference modules: .toc3d_eva_vit, .eva_vit, .toc3d_utils,
    .eva_utils, .misc, .posenc, .BACKBONES."""
    if _LOADED:
        return types.SimpleNamespace(**_LOADED)
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REF}; the oracle harness only runs in the build container")
    sys.dont_write_bytecode = True  # the reference tree is read-only

    backbones = _Registry()
    necks = _Registry()
    _mod("mmdet")
    _mod("mmdet.models", NECKS=necks)
    _mod("mmdet.models.builder", BACKBONES=backbones, NECKS=necks)       # toc3d_eva_vit.py:6, eva_vit.py:8
    _mod("mmdet3d")
    _mod("mmdet3d.models")
    _mod("mmdet3d.models.builder", build_loss=lambda cfg: None)          # toc3d_eva_vit.py:7
    _mod("mmdet.core", bbox_xyxy_to_cxcywh=None)                         # models/utils/misc.py:4
    _mod("mmdet.models.utils")
    _mod("mmdet.models.utils.transformer", inverse_sigmoid=None)         # models/utils/misc.py:5

    class ShapeSpec:                                                      # eva_utils.py:10
        def __init__(self, channels=None, stride=None):
            self.channels, self.stride = channels, stride

    class CNNBlockBase(nn.Module):                                        # eva_vit.py:9 (base of an unused class)
        def __init__(self, i, o, s):
            super().__init__()

    _mod("detectron2")
    _mod("detectron2.layers", ShapeSpec=ShapeSpec, CNNBlockBase=CNNBlockBase, Conv2d=nn.Conv2d, get_norm=None)
    _mod("detectron2.modeling")
    _mod("detectron2.modeling.backbone")
    _mod("detectron2.modeling.backbone.fpn", _assert_strides_are_log2_contiguous=None)   # eva_vit.py:10
    _mod("fvcore")
    _mod("fvcore.nn")
    sys.modules["fvcore.nn"].weight_init = _mod("fvcore.nn.weight_init")                 # eva_vit.py:4

    class DropPath(nn.Module):                                            # eva_vit.py:224 (identity in eval)
        def __init__(self, p):
            super().__init__()

        def forward(self, x):
            assert not self.training
            return x

    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=DropPath)
    _mod("fairscale")
    _mod("fairscale.nn")
    _mod("fairscale.nn.checkpoint", checkpoint_wrapper=lambda b: b)       # toc3d_eva_vit.py:203

    # mmcv pieces used by necks/cp_fpn.py:9-12 (ConvModule with no norm/act is a bare Conv2d under `.conv`)
    class ConvModule(nn.Module):
        def __init__(self, cin, cout, k, stride=1, padding=0, conv_cfg=None, norm_cfg=None, act_cfg=None, inplace=False):
            super().__init__()
            assert norm_cfg is None and act_cfg is None
            self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding)

        def forward(self, x):
            return self.conv(x)

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()

    def auto_fp16(*a, **k):
        return lambda f: f

    _mod("mmcv")
    _mod("mmcv.cnn", ConvModule=ConvModule)
    _mod("mmcv.runner", BaseModule=BaseModule, auto_fp16=auto_fp16)

    # bypass projects/mmdet3d_plugin/__init__.py (imports datasets/heads that need the real mmdet3d)
    for name, path in [
        ("projects", "projects"),
        ("projects.mmdet3d_plugin", "projects/mmdet3d_plugin"),
        ("projects.mmdet3d_plugin.models", "projects/mmdet3d_plugin/models"),
        ("projects.mmdet3d_plugin.models.backbones", "projects/mmdet3d_plugin/models/backbones"),
        ("projects.mmdet3d_plugin.models.necks", "projects/mmdet3d_plugin/models/necks"),
        ("projects.mmdet3d_plugin.models.utils", "projects/mmdet3d_plugin/models/utils"),
    ]:
        _mod(name).__path__ = [f"{REF}/{path}"]

    imp = importlib.import_module
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        _LOADED["toc3d_eva_vit"] = imp("projects.mmdet3d_plugin.models.backbones.toc3d_eva_vit")
        _LOADED["eva_vit"] = imp("projects.mmdet3d_plugin.models.backbones.eva_vit")
        _LOADED["toc3d_utils"] = imp("projects.mmdet3d_plugin.models.backbones.toc3d_utils")
        _LOADED["eva_utils"] = imp("projects.mmdet3d_plugin.models.backbones.eva_utils")
        _LOADED["misc"] = imp("projects.mmdet3d_plugin.models.utils.misc")
        _LOADED["posenc"] = imp("projects.mmdet3d_plugin.models.utils.positional_encoding")
        _LOADED["cp_fpn"] = imp("projects.mmdet3d_plugin.models.necks.cp_fpn")
    _LOADED["BACKBONES"] = backbones
    _LOADED["NECKS"] = necks
    return types.SimpleNamespace(**_LOADED)


@contextlib.contextmanager
def deterministic_reference(gumbel_noise=None):
    """Patches torch.sort (stable) and F.gumbel_softmax (injected noise) while the reference runs.

    ``gumbel_noise``: list of tensors consumed in call order by calls whose logits have a
    last dim of 2 (the 3 image-level calls); window-level calls (last dim 1) get zeros -- their
    result is a softmax over a size-1 dim (always 1) and is discarded by the reference
    (``toc3d_eva_vit.py:419`` ignores the 5th return).
    """
    orig_sort = torch.sort
    orig_gs = F.gumbel_softmax
    queue = list(gumbel_noise) if gumbel_noise is not None else None
    calls = {"n_img": 0, "n_win": 0}

    def stable_sort(input, dim=-1, descending=False, stable=False, **kw):
        return orig_sort(input, dim=dim, descending=descending, stable=True, **kw)

    def injected_gumbel(logits, tau=1, hard=False, eps=1e-10, dim=-1):
        assert tau == 1 and not hard
        if logits.shape[-1] == 1:
            calls["n_win"] += 1
            return torch.softmax(logits, dim=dim)
        calls["n_img"] += 1
        if queue is None:
            g = torch.zeros_like(logits)
        else:
            g = queue.pop(0).to(logits.dtype).reshape(logits.shape)
        return torch.softmax(logits + g, dim=dim)

    torch.sort = stable_sort
    F.gumbel_softmax = injected_gumbel
    try:
        yield calls
    finally:
        torch.sort = orig_sort
        F.gumbel_softmax = orig_gs


def build_reference_toc3d(cfg: dict):
    """Instantiate the reference ToC3DEVAViT from a config dict (token_selection_loss dropped), eval mode."""
    ref = load_reference()
    kw = dict(cfg)
    kw["token_selection_loss"] = None
    kw.pop("type", None)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        m = ref.toc3d_eva_vit.ToC3DEVAViT(**kw)
    return m.eval()


def build_reference_eva(cfg: dict):
    ref = load_reference()
    kw = {k: v for k, v in cfg.items() if k not in (
        "type", "rope_acc", "pruning_loc", "pruning_score_type", "score_mask", "pruning_attn_scale",
        "pruning_num_queries", "accelerate_global", "token_ratio", "use_represent_tokens", "pc_range",
        "token_selection_loss")}
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        m = ref.eva_vit.EVA_ViT(**kw)
    return m.eval()


def build_reference_cpfpn(cfg: dict):
    ref = load_reference()
    kw = {k: v for k, v in cfg.items() if k != "type"}
    return ref.cp_fpn.CPFPN(**kw).eval()


# ---------------------------------------------------------------------------------------------------
# StreamPETRHead temporal memory: only the three memory methods are exercised; everything
# the module imports at top level is stubbed (dense_heads/streampetr_head.py:12-31).
_HEAD = {}


def load_reference_head():
    """Returns the reference ``StreamPETRHead`` class (methods used unbound on a plain object holding the memory state)."""
    if _HEAD:
        return _HEAD["cls"]
    load_reference()
    heads = _Registry()
    sys.modules["mmcv.cnn"].Linear = nn.Linear
    sys.modules["mmcv.cnn"].bias_init_with_prob = lambda p: 0.0
    sys.modules["mmcv.runner"].force_fp32 = lambda *a, **k: (lambda f: f)
    core = sys.modules["mmdet.core"]
    core.build_assigner = core.build_sampler = core.multi_apply = core.reduce_mean = None
    sys.modules["mmdet.models.utils"].build_transformer = None
    sys.modules["mmdet.models.utils"].NormedLinear = nn.Linear
    sys.modules["mmdet.models"].HEADS = heads

    class AnchorFreeHead(nn.Module):
        pass

    _mod("mmdet.models.dense_heads")
    _mod("mmdet.models.dense_heads.anchor_free_head", AnchorFreeHead=AnchorFreeHead)
    _mod("mmdet3d.core")
    _mod("mmdet3d.core.bbox")
    _mod("mmdet3d.core.bbox.coders", build_bbox_coder=None)
    sys.modules["mmdet3d.models"].build_loss = lambda cfg: None
    for name, path in [("projects.mmdet3d_plugin.core", "projects/mmdet3d_plugin/core"),
                       ("projects.mmdet3d_plugin.core.bbox", "projects/mmdet3d_plugin/core/bbox"),
                       ("projects.mmdet3d_plugin.models.dense_heads", "projects/mmdet3d_plugin/models/dense_heads")]:
        _mod(name).__path__ = [f"{REF}/{path}"]
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        mod = importlib.import_module("projects.mmdet3d_plugin.models.dense_heads.streampetr_head")
    _HEAD["cls"] = mod.StreamPETRHead
    return _HEAD["cls"]


class ReferenceMemory:
    """Plain holder for the attributes ``StreamPETRHead.{reset,pre_update,post_update}_memory`` touch
    (dense_heads/streampetr_head.py:315-377), driven through the reference's own (unbound) methods."""

    def __init__(self, memory_len, topk_proposals, num_propagated, embed_dims, pc_range, pseudo_reference_points):
        self.memory_len, self.topk_proposals, self.num_propagated, self.embed_dims = memory_len, topk_proposals, num_propagated, embed_dims
        self.pc_range = torch.tensor(pc_range, dtype=torch.float32)
        self.pseudo_reference_points = types.SimpleNamespace(weight=pseudo_reference_points)
        self.training = False
        self._cls = load_reference_head()
        self._cls.reset_memory(self)

    def pre_update_memory(self, data):
        self._cls.pre_update_memory(self, data)

    def post_update_memory(self, data, rec_ego_pose, all_cls_scores, all_bbox_preds, outs_dec):
        orig = torch.topk

        def stable_topk(input, k, dim=-1, largest=True, sorted=True):       # tie rule pinned like torch.sort elsewhere
            v, i = torch.sort(input, dim=dim, descending=largest, stable=True)
            return v.narrow(dim, 0, k), i.narrow(dim, 0, k)

        torch.topk = stable_topk
        try:
            self._cls.post_update_memory(self, data, rec_ego_pose, all_cls_scores, all_bbox_preds, outs_dec, None)
        finally:
            torch.topk = orig
