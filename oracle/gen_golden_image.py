"""Generates tests/golden/image_norm.npz by driving the reference's OWN pipeline classes -- ``NormalizeMultiviewImage.__call__``
(``datasets/pipelines/transform_3d.py:87-100``) and ``PadMultiViewImage._pad_img`` (``:38-50``), imported where they lie under /root/reference -- with the
configs' arguments (``projects/configs/ToC3D/ToC3D_faster.py:13-14,206-214``: mean / std / to_rgb, size_divisor = 32).  Build container only.

WHAT IS PINNED AND WHAT IS NOT.  The classes (argument handling, float32 mean / std, the order normalise -> pad, per-view lists, the pad geometry's caller) are
the reference's code, executed.  The two mmcv functions they call, ``mmcv.imnormalize`` and ``mmcv.impad_to_multiple`` (mmcv-full 1.6.0, README.md:52; OpenCV
behind them), are NOT installed here and are stood in for by the stubs below, written from the published sources (mmcv/image/photometric.py, geometric.py;
OpenCV arithm_op on a CV_32F matrix with a double scalar works in CV_32F).  So this fixture pins oracle/image_oracle.py against the reference's classes; the
stubs' arithmetic is the part that stays "parity unpinned" upstream, and the test that reads this file says so.
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FILE = "/root/reference/projects/mmdet3d_plugin/datasets/pipelines/transform_3d.py"


# ---- stand-ins for the absent third-party functions (published definitions) ---------------------------------------------------------
def _cv_subtract_scalar(img32, scalar64):           # cv2.subtract(CV_32FC3, double scalar, dst CV_32F): the scalar is converted to the matrix depth
    return (img32 - scalar64.astype(np.float32)).astype(np.float32)


def _cv_multiply_scalar(img32, scalar64):           # cv2.multiply likewise
    return (img32 * scalar64.astype(np.float32)).astype(np.float32)


def imnormalize(img, mean, std, to_rgb=True):       # mmcv/image/photometric.py: imnormalize -> imnormalize_
    img = img.copy().astype(np.float32)
    assert img.dtype != np.uint8
    mean = np.float64(mean.reshape(1, -1))
    stdinv = 1 / np.float64(std.reshape(1, -1))
    if to_rgb:
        img = np.ascontiguousarray(img[..., ::-1])  # cv2.cvtColor(img, cv2.COLOR_BGR2RGB, img)
    img = _cv_subtract_scalar(img, mean)
    return _cv_multiply_scalar(img, stdinv)


def impad(img, *, shape=None, padding=None, pad_val=0, padding_mode="constant"):      # mmcv/image/geometric.py: cv2.copyMakeBorder(..., BORDER_CONSTANT, value=pad_val)
    assert padding is None and padding_mode == "constant"
    out = np.full((shape[0], shape[1]) + img.shape[2:], pad_val, dtype=img.dtype)
    out[: img.shape[0], : img.shape[1]] = img
    return out


def impad_to_multiple(img, divisor, pad_val=0):     # mmcv/image/geometric.py
    pad_h = int(np.ceil(img.shape[0] / divisor)) * divisor
    pad_w = int(np.ceil(img.shape[1] / divisor)) * divisor
    return impad(img, shape=(pad_h, pad_w), pad_val=pad_val)


def load_reference_transforms():
    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls
    sys.dont_write_bytecode = True
    for name, kw in (("mmcv", dict(imnormalize=imnormalize, impad=impad, impad_to_multiple=impad_to_multiple)), ("mmdet", {}), ("mmdet.datasets", {}),
                     ("mmdet.datasets.builder", dict(PIPELINES=_Registry()))):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("ref_transform_3d", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    T = load_reference_transforms()
    rng = np.random.Generator(np.random.Philox(key=20240905))
    out = {}
    cfg = dict(mean=[103.530, 116.280, 123.675], std=[57.375, 57.120, 58.395])             # ToC3D_faster.py:13-14
    cases = {"a": (3, 45, 70, False), "b": (2, 64, 96, True), "c": (1, 33, 47, True), "d": (1, 40, 800, False)}     # (views, H, W, to_rgb); d: a full-width strip
    for tag, (V, H, W, to_rgb) in cases.items():
        u8 = rng.integers(0, 256, size=(V, H, W, 3), dtype=np.uint8)
        u8[0, 0, 0] = (0, 255, 128)
        res = {"img": [v.astype(np.float32) for v in u8]}                                   # LoadMultiViewImageFromFiles(to_float32=True): integer-valued float32 HWC BGR
        res = T.NormalizeMultiviewImage(cfg["mean"], cfg["std"], to_rgb=to_rgb)(res)       # the order of the test pipeline, ToC3D_faster.py:206-214
        res = T.PadMultiViewImage(size_divisor=32)(res)
        assert res["pad_size_divisor"] == 32 and all(s == (H, W, 3) for s in res["img_shape"])
        bundle = np.ascontiguousarray(np.stack(res["img"], 0).transpose(0, 3, 1, 2))        # DefaultFormatBundle image branch (mmdet3d formating.py:42-47): HWC -> CHW, views stacked
        out[f"{tag}_u8"], out[f"{tag}_to_rgb"], out[f"{tag}_expected"] = u8, np.array(to_rgb), bundle.astype(np.float32)
    out["mean"], out["std"], out["size_divisor"] = np.array(cfg["mean"], np.float64), np.array(cfg["std"], np.float64), np.array(32)
    out["pinned"] = np.array("classes: /root/reference transform_3d.py NormalizeMultiviewImage / PadMultiViewImage (executed); "
                             "mmcv.imnormalize / impad_to_multiple: stand-ins from the published mmcv 1.6.0 / OpenCV definitions (oracle/gen_golden_image.py) -- the unpinned part")
    path = os.path.join(ROOT, "tests", "golden", "image_norm.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
