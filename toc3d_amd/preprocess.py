"""Image side of the patch embedding on the GPU.

The reference normalises, pads and transposes every camera image on the host and ships float32 NCHW to the GPU
(``NormalizeMultiviewImage`` / ``PadMultiViewImage``, ``datasets/pipelines/transform_3d.py:87-100,38-50``;
``DefaultFormatBundle``, ``mmdet3d/datasets/pipelines/formating.py:42-47``).  The pixels are integer valued at that point
(``LoadMultiViewImageFromFiles(to_float32=True)`` + PIL resize/crop of uint8 data), so the same tensor can be produced on
the device from the uint8 HWC images: a quarter of the host-to-device bytes (4.6 MB instead of 18.4 MB per 6-view frame).

``prepare_images`` materialises that float tensor (drop-in for the three pipeline steps); a backbone built with
``img_norm_cfg=...`` skips even that and reads the uint8 images straight in its patch-embedding im2col
(``toc3d_im2col_patches_u8``), with bit-identical results.
"""
from __future__ import annotations

from typing import Sequence

import torch

from . import lib


def prepare_images(img_u8: torch.Tensor, mean: Sequence[float], std: Sequence[float], to_rgb: bool = True,
                   size_divisor: int = 32) -> torch.Tensor:
    """uint8 HWC (V, H, W, 3) on the GPU -> float32 NCHW (V, 3, Hp, Wp), Hp / Wp = H / W rounded up to ``size_divisor``."""
    if not isinstance(img_u8, torch.Tensor) or not img_u8.is_cuda:
        raise RuntimeError("toc3d_amd.preprocess: images must be CUDA/HIP tensors -- the HIP extension is the only compute path")
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[3] != 3:
        raise ValueError(f"expected uint8 HWC images (V, H, W, 3), got {img_u8.dtype} {tuple(img_u8.shape)}")
    if size_divisor % 4:
        raise ValueError("size_divisor must be a multiple of 4")
    img_u8 = img_u8.contiguous()
    V, H, W, _ = img_u8.shape
    Hp, Wp = -(-H // size_divisor) * size_divisor, -(-W // size_divisor) * size_divisor
    out = torch.empty(V, 3, Hp, Wp, dtype=torch.float32, device=img_u8.device)
    m = torch.tensor(list(mean), dtype=torch.float32)
    s = torch.tensor(list(std), dtype=torch.float32)
    lib.call("toc3d_normalize_images", img_u8, V, H, W, m, s, int(bool(to_rgb)), out, Hp, Wp, lib.stream_ptr())
    return out
