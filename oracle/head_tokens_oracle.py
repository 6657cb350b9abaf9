"""TEST INFRASTRUCTURE ONLY -- CPU (eager torch) restatement of the token side of StreamPETRHead.forward: what consumes the
neck's features right behind the hot path (second half).

Follows ``dense_heads/streampetr_head.py``: ``position_embeding`` :378-422 (frustum points of every image token through
``lidar2img^-1``, normalised by ``position_range``, ``inverse_sigmoid``, ``position_encoder`` :262-266), the ``cone`` vector for the
spatial alignment :419-420, and ``forward`` :627-639 (``memory_embed`` :268-272, ``spatial_alignment`` = ``MLN(8)``
``models/utils/misc.py:154-188``, ``featurized_pe`` = ``SELayer_Linear`` ``misc.py:139-151``); token centres from ``locations``
``misc.py:59-82`` / ``Petr3D.prepare_location`` ``detectors/petr3d.py:311-316``.  ``inverse_sigmoid`` lives in mmdet 2.28
(``mmdet/models/utils/transformer.py``, absent here): restated from the published source (clamp to [0, 1], eps = 1e-5).

Pinned: ``oracle/gen_golden_head.py`` drives the reference's own ``position_embeding`` (unbound, on an object holding the same
sub-modules) and module classes (``MLN``, ``SELayer_Linear``) and commits ``tests/golden/head_tokens.npz``.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def inverse_sigmoid(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def coords_d(position_range, depth_num: int = 64, depth_start: float = 1.0, lid: bool = True) -> torch.Tensor:
    """streampetr_head.py:221-232."""
    pr = torch.tensor(position_range, dtype=torch.float32)
    index = torch.arange(0, depth_num, 1).float()
    if lid:
        bin_size = (pr[3] - depth_start) / (depth_num * (1 + depth_num))
        return depth_start + bin_size * index * (index + 1)
    return depth_start + (pr[3] - depth_start) / depth_num * index


def locations(h: int, w: int, stride: int, pad_h: int, pad_w: int) -> torch.Tensor:
    """misc.py:59-82 -> (h, w, 2) normalised token centres (x, y)."""
    sx = (torch.arange(0, stride * w, step=stride, dtype=torch.float32) + stride // 2) / pad_w
    sy = (torch.arange(0, h * stride, step=stride, dtype=torch.float32) + stride // 2) / pad_h
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    return torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1).reshape(h, w, 2)


def linear(sd, pre, x):
    return F.linear(x, sd[pre + "weight"], sd[pre + "bias"])


def position_embedding(sd: Dict[str, torch.Tensor], cfg: dict, intrinsics, lidar2img, h: int, w: int, pad_h: int, pad_w: int):
    """:378-422 with topk_indexes = None.  intrinsics / lidar2img (B, N, 4, 4).  Returns pos (B, N*h*w, C), cone (B, N*h*w, 8)."""
    eps = 1e-5
    B, N = intrinsics.shape[:2]
    cd = coords_d(cfg["position_range"], cfg["depth_num"], cfg["depth_start"], cfg["LID"])
    D = cd.shape[0]
    LEN = N * h * w
    intr = torch.stack([intrinsics[..., 0, 0], intrinsics[..., 1, 1]], dim=-1)            # (B, N, 2)
    intr = (torch.abs(intr) / 1e3).repeat(1, h * w, 1).view(B, -1, 2)                       # reference's repeat order (:385)
    centers = locations(h, w, cfg["stride"], pad_h, pad_w)[None].repeat(B * N, 1, 1, 1).clone()
    centers[..., 0] = centers[..., 0] * pad_w
    centers[..., 1] = centers[..., 1] * pad_h
    centers = centers.view(B, LEN, 1, 2).repeat(1, 1, D, 1)
    coords = torch.cat([centers, cd.view(1, 1, D, 1).repeat(B, LEN, 1, 1)], dim=-1)
    coords = torch.cat((coords, torch.ones_like(coords[..., :1])), -1)
    coords[..., :2] = coords[..., :2] * torch.maximum(coords[..., 2:3], torch.ones_like(coords[..., 2:3]) * eps)
    img2lidars = torch.inverse(lidar2img).view(B * N, 1, 1, 4, 4).repeat(1, h * w, D, 1, 1).view(B, LEN, D, 4, 4)
    coords3d = torch.matmul(img2lidars, coords.unsqueeze(-1)).squeeze(-1)[..., :3]
    pr = torch.tensor(cfg["position_range"], dtype=torch.float32)
    coords3d = (coords3d - pr[0:3]) / (pr[3:6] - pr[0:3])
    coords3d = coords3d.reshape(B, -1, D * 3)
    pos = inverse_sigmoid(coords3d)
    pos = linear(sd, "position_encoder.2.", F.relu(linear(sd, "position_encoder.0.", pos)))
    cone = torch.cat([intr, coords3d[..., -3:], coords3d[..., -90:-87]], dim=-1)
    return pos, cone


def token_embeddings(sd, cfg, feats, intrinsics, lidar2img, pad_h: int, pad_w: int):
    """forward :627-639: feats (B, N, C, h, w) neck level-0 -> (memory, pos_embed), both (B, N*h*w, embed_dims)."""
    B, N, C, h, w = feats.shape
    memory = feats.permute(0, 1, 3, 4, 2).reshape(B, N * h * w, C)
    pos, cone = position_embedding(sd, cfg, intrinsics, lidar2img, h, w, pad_h, pad_w)
    memory = linear(sd, "memory_embed.2.", F.relu(linear(sd, "memory_embed.0.", memory)))
    # MLN (misc.py:181-188): LayerNorm without affine, scale / shift from the reduced cone vector
    x = F.layer_norm(memory, (memory.shape[-1],), None, None, 1e-5)
    c = F.relu(linear(sd, "spatial_alignment.reduce.0.", cone))
    memory = linear(sd, "spatial_alignment.gamma.", c) * x + linear(sd, "spatial_alignment.beta.", c)
    # SELayer_Linear (misc.py:147-151)
    se = linear(sd, "featurized_pe.conv_expand.", F.relu(linear(sd, "featurized_pe.conv_reduce.", memory)))
    return memory, pos * torch.sigmoid(se)
