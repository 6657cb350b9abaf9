"""Packed-weight cache + checkpoint converter.

The reference loads a ``.pth`` with ``load_checkpoint(model, ckpt, map_location='cpu')`` (``tools/test.py:207``) on every start.
Here the weights additionally go through a device-side packing step (fused q|k|v, interleaved w1|w2, 2730 -> 2752 padding,
kept-pad q|k|v rows, motion-query tables, bicubic-resized abs-pos per resolution).
``save_packed`` writes exactly what the kernels consume as one ``safetensors`` file next to the checkpoint;
``load_packed`` restores it without the state dict and without re-packing.

File: safetensors; tensor names are the paths inside the packed dict (``blocks.7.wqkv``, ``scorers.1.w_in``,
``pos.20x50``); non-tensor leaves (ints, floats) and the model fingerprint live in the metadata as JSON.
"""
from __future__ import annotations

import json
from typing import Any, Dict, Tuple

import torch

FORMAT = "toc3d_amd.packed/1"


def _flatten(node: Any, prefix: str, tensors: Dict[str, torch.Tensor], scalars: Dict[str, Any]):
    if isinstance(node, torch.Tensor):
        tensors[prefix] = node.detach().contiguous().cpu()
    elif isinstance(node, dict):
        for k, v in node.items():
            key = "x".join(str(i) for i in k) if isinstance(k, tuple) else str(k)
            _flatten(v, f"{prefix}.{key}" if prefix else key, tensors, scalars)
    elif isinstance(node, (list, tuple)):
        scalars[prefix + ".#len"] = len(node)
        for i, v in enumerate(node):
            _flatten(v, f"{prefix}.{i}", tensors, scalars)
    elif node is None or isinstance(node, (bool, int, float, str)):
        scalars[prefix] = node
    else:
        raise TypeError(f"cannot serialise packed entry {prefix}: {type(node)}")


def _insert(root: dict, path: str, value: Any):
    parts = path.split(".")
    node = root
    for p in parts[:-1]:
        node = node.setdefault(p, {})
    node[parts[-1]] = value


def _listify(node: Any):
    if not isinstance(node, dict):
        return node
    if "#len" in node:
        return [_listify(node[str(i)]) for i in range(node["#len"])]
    return {k: _listify(v) for k, v in node.items()}


def fingerprint(model) -> Dict[str, Any]:
    """What must agree between the model a file was written from and the model it is loaded into."""
    from . import lib
    fp = dict(cls=type(model).__name__, precision=model.precision, embed_dim=model.embed_dim, depth=model.depth, num_heads=model.num_heads,
              hidden_dim=model.hidden_dim, patch_size=model.patch_size, window_size=model.window_size,
              global_window_size=model.global_window_size, global_attn_indexes=list(model.global_attn_indexes),
              in_chans=model.in_chans, pretrain_use_cls_token=bool(model.pretrain_use_cls_token),
              # everything that shapes or gives meaning to a packed buffer: the abs-pos source grid (None = use_abs_pos False), the RoPE
              # table sides (img_size / patch for the global blocks), the scorer scale and the point-cloud range baked into the motion tables
              pos_embed_rows=None if model.pos_embed is None else int(model.pos_embed.shape[1]),
              rope_sides=[int(model.rope_win.freqs_cos.shape[0]), int(model.rope_glb.freqs_cos.shape[0])],
              fold_ffn_ln=bool(getattr(model, "fold_ffn_ln", False)),      # w3 packed gamma-scaled + c1 / c2 instead of w3 / b3
              fold_norm2=bool(getattr(model, "fold_norm2", False)),        # w12 packed gamma2-scaled + c1_12 / c2_12
              attn_rot=bool(getattr(model, "attn_rot", False)),            # compact RoPE tables + per-slot rotated pad rows (pad_rot)
              x3_planes=bool(getattr(model, "x3_planes", False)),          # fp32x3: GEMM weights stored as (hi, lo) bf16 planes
              abi=int(lib.load().toc3d_abi_version()))
    for k in ("pruning_loc", "token_ratio", "pruning_num_queries", "accelerate_global", "pruning_attn_scale"):
        if hasattr(model, k):
            v = getattr(model, k)
            fp[k] = list(v) if isinstance(v, (list, tuple)) else v
    if hasattr(model, "score_predictor") and len(model.score_predictor):
        fp["pc_range"] = [round(float(v), 6) for v in model.score_predictor[0].pc_range.detach().cpu().tolist()]
    return fp


def save_packed(model, path: str):
    """Pack (if not done yet) and write the packed weights of ``model`` (a toc3d_amd backbone on the GPU) to ``path``."""
    from safetensors.torch import save_file
    if model._packed is None:
        model._packed = model._pack() if hasattr(model, "_pack") else model._pack_common()
    P = {k: v for k, v in model._packed.items() if k != "dev"}
    tensors: Dict[str, torch.Tensor] = {}
    scalars: Dict[str, Any] = {}
    _flatten(P, "", tensors, scalars)
    meta = {"format": FORMAT, "fingerprint": json.dumps(fingerprint(model)), "scalars": json.dumps(scalars)}
    save_file(tensors, path, metadata=meta)


def load_packed(model, path: str):
    """Restore packed weights written by ``save_packed`` into ``model`` (already on the GPU); no state dict needed.
    Raises if the file was written for a different architecture / precision / ABI."""
    from safetensors import safe_open
    dev = model.patch_embed.proj.weight.device
    if dev.type != "cuda":
        raise RuntimeError("load_packed: move the module to the GPU first (packed weights are device buffers; no CPU path)")
    root: dict = {}
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        if meta.get("format") != FORMAT:
            raise ValueError(f"{path}: not a {FORMAT} file")
        want, have = fingerprint(model), json.loads(meta["fingerprint"])
        if want != have:
            diff = {k: (have.get(k), want.get(k)) for k in set(want) | set(have) if have.get(k) != want.get(k)}
            raise ValueError(f"{path} was packed for a different model (file, model): {diff}")
        for k, v in json.loads(meta["scalars"]).items():
            _insert(root, k, v)
        for k in f.keys():
            _insert(root, k, f.get_tensor(k).to(dev))
    P = _listify(root)
    P["pos"] = {tuple(int(i) for i in k.split("x")): v for k, v in P.get("pos", {}).items()}
    P["dev"] = dev
    model._packed = P
    model._plans = {}                      # recorded launch plans point into the previous packed buffers


def convert_checkpoint(ckpt_path: str, cfg: dict, out_path: str, prefix: str = "img_backbone.", device: str = "cuda",
                       trust_pickle: bool = False):
    """``.pth`` of the reference detector (``tools/test.py:207``; keys ``img_backbone.*``) -> packed file for ``cfg``
    (a backbone config dict, ``type`` included).  Returns the backbone it built.  The checkpoint is read with
    ``weights_only=True`` (tensors and plain containers only); ``trust_pickle=True`` opts into full unpickling for checkpoints
    whose ``meta`` holds arbitrary objects -- only for files you trust, unpickling can run code."""
    from .registry import build_backbone
    ck = torch.load(ckpt_path, map_location="cpu", weights_only=not trust_pickle)
    sd = ck.get("state_dict", ck)
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else dict(sd)
    if not sub:
        raise KeyError(f"no '{prefix}*' keys in {ckpt_path}")
    model = build_backbone(cfg)
    model.load_state_dict(sub, strict=True)
    model = model.to(device).eval()
    save_packed(model, out_path)
    return model
