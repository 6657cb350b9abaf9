"""TEST INFRASTRUCTURE ONLY -- CPU (numpy, f32) restatement of the image side of the patch embedding: the step
directly upstream of the backbone in the reference's test pipeline.

Follows, in order (``projects/configs/ToC3D/ToC3D_faster.py:206-214``):
  * ``NormalizeMultiviewImage.__call__``  (``datasets/pipelines/transform_3d.py:87-100``) -> ``mmcv.imnormalize``
  * ``PadMultiViewImage._pad_img``        (``datasets/pipelines/transform_3d.py:38-50``)  -> ``mmcv.impad_to_multiple``
  * ``DefaultFormatBundle`` image branch  (``mmdetection3d/mmdet3d/datasets/pipelines/formating.py:42-47``): HWC -> CHW, stack views

PINNED AGAINST THE REFERENCE'S CLASSES since round 5: tests/golden/image_norm.npz is the output of the reference's own ``NormalizeMultiviewImage`` /
``PadMultiViewImage`` (executed from /root/reference by oracle/gen_golden_image.py) and this file reproduces it bit for bit
(tests/test_cpu_oracle_golden.py::test_image_oracle_against_the_reference_pipeline_classes).  What stays PARITY UNPINNED is the arithmetic of the two mmcv
functions those classes call, which the generator had to stand in for from their published definitions: the arithmetic lives in mmcv-full 1.6.0 (``README.md:52``) and OpenCV, neither of which
is vendored in /root/reference nor installed here, and the reference holds no test or golden vector for it.  Restated
from the published sources:
  mmcv/image/photometric.py ``imnormalize_``:  ``mean = float64(mean); stdinv = 1 / float64(std);
      if to_rgb: cv2.cvtColor(img, COLOR_BGR2RGB, img); cv2.subtract(img, mean, img); cv2.multiply(img, stdinv, img)``
  OpenCV ``arithm_op`` with a CV_32F matrix and a double scalar works in CV_32F (the scalar is converted to float),
  so each step rounds to f32:  out = fl32(fl32(x - fl32(mean)) * fl32(stdinv)).
  mmcv/image/geometric.py ``impad_to_multiple``: pad bottom / right with ``pad_val`` to the next multiple of the divisor
  (applied AFTER the normalisation, so padded pixels are exactly 0.0 in the normalised image).
The images reach the pipeline as integer-valued float32 HWC BGR arrays (``LoadMultiViewImageFromFiles(to_float32=True)``
then PIL resize/crop of uint8 data in ``ResizeCropFlipRotImage``), which is why a uint8 boundary loses nothing.
"""
from __future__ import annotations

import numpy as np


def imnormalize(img: np.ndarray, mean, std, to_rgb: bool) -> np.ndarray:
    """img (H, W, 3) uint8 or integer-valued f32 -> f32 (H, W, 3).  transform_3d.py:96-97."""
    x = img.astype(np.float32)
    m32 = np.float64(np.asarray(mean, np.float32)).astype(np.float32)
    s32 = (1.0 / np.float64(np.asarray(std, np.float32))).astype(np.float32)
    if to_rgb:
        x = x[..., ::-1]
    return ((x - m32[None, None, :]).astype(np.float32) * s32[None, None, :]).astype(np.float32)


def impad_to_multiple(img: np.ndarray, divisor: int, pad_val: float = 0.0) -> np.ndarray:
    """transform_3d.py:43-45."""
    H, W = img.shape[:2]
    Hp, Wp = -(-H // divisor) * divisor, -(-W // divisor) * divisor
    out = np.full((Hp, Wp) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:H, :W] = img
    return out


def prepare_images(imgs_u8: np.ndarray, mean, std, to_rgb: bool, size_divisor: int) -> np.ndarray:
    """(V, H, W, 3) uint8 -> (V, 3, Hp, Wp) f32: what ``ToC3DEVAViT.forward`` receives as ``x`` (petr3d.py:139-141)."""
    views = [impad_to_multiple(imnormalize(v, mean, std, to_rgb), size_divisor) for v in imgs_u8]
    return np.ascontiguousarray(np.stack([v.transpose(2, 0, 1) for v in views], axis=0))
