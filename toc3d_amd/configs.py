"""Backbone configuration dicts, written in the reference's own config vocabulary.

The dicts below carry the same keys as ``img_backbone=dict(...)`` in the reference configs
(``projects/configs/ToC3D/ToC3D_faster.py:41-69``, ``ToC3D_fast.py:64``) so that a reference
config can be passed verbatim to :class:`toc3d_amd.backbone.ToC3DEVAViT`.
"""
from __future__ import annotations

import copy

POINT_CLOUD_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]   # ToC3D_faster.py:11

_VITL = dict(
    img_size=320, patch_size=16, window_size=16, global_window_size=20, in_chans=3,
    embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4 * 2 / 3,
    global_attn_indexes=(2, 5, 8, 11, 14, 17, 20, 23), qkv_bias=True, drop_path_rate=0.3,
    use_act_checkpoint=True, xattn=False, use_checkpoint=False, rope=True,
)

TOC3D_FASTER = dict(
    type="ToC3DEVAViT", **_VITL, rope_acc=True, pc_range=POINT_CLOUD_RANGE, pruning_num_queries=64,
    pruning_loc=[6, 12, 18], accelerate_global=True, token_ratio=[0.5, 0.4, 0.3],
)
TOC3D_FAST = dict(TOC3D_FASTER, token_ratio=[0.7, 0.5, 0.5])          # ToC3D_fast.py:64
EVA_DENSE = dict(type="EVA_ViT", **_VITL)                               # StreamPETR baseline backbone

# small configuration used by the parity tests (SURVEY.md 8c "F-tiny-e2e"): same structure,
# head dim 64 kept (2 heads x 64), 12 blocks, scorers before blocks 3/6/9.
_TINY = dict(
    img_size=320, patch_size=16, window_size=16, global_window_size=20, in_chans=3,
    embed_dim=128, depth=12, num_heads=2, mlp_ratio=4 * 2 / 3,
    global_attn_indexes=(2, 5, 8, 11), qkv_bias=True, drop_path_rate=0.0,
    use_act_checkpoint=False, xattn=False, use_checkpoint=False, rope=True,
)
TOC3D_TINY = dict(
    type="ToC3DEVAViT", **_TINY, rope_acc=True, pc_range=POINT_CLOUD_RANGE, pruning_num_queries=64,
    pruning_loc=[3, 6, 9], accelerate_global=True, token_ratio=[0.5, 0.4, 0.3],
)
EVA_TINY = dict(type="EVA_ViT", **_TINY)

CPFPN_CFG = dict(type="CPFPN", in_channels=[1024], out_channels=256, num_outs=2)   # ToC3D_faster.py:70-74
CPFPN_TINY = dict(type="CPFPN", in_channels=[128], out_channels=32, num_outs=2)

NAMED = {
    "toc3d_faster": TOC3D_FASTER,
    "toc3d_fast": TOC3D_FAST,
    "eva_dense": EVA_DENSE,
    "toc3d_tiny": TOC3D_TINY,
    "eva_tiny": EVA_TINY,
}


def get(name: str) -> dict:
    return copy.deepcopy(NAMED[name])
