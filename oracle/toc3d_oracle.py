"""TEST INFRASTRUCTURE ONLY -- CPU (eager torch, fp32) restatement of the reference hot path.

This is the *checker* for the HIP path, never the thing shipped or measured (except as the
``cpu_baseline`` leg of ``bench.py``).  It is a functional re-derivation of the reference
algorithm over a plain state-dict; every function cites the reference lines it follows
(paths relative to ``/root/reference/projects/mmdet3d_plugin/models/``).

Pinning: ``oracle/gen_golden.py`` runs the real reference (imported with stub registries, see
``oracle/ref_harness.py``) on the same seeded weights/inputs and commits small fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against them.  The
reference itself has no tests/golden vectors for this path, so those
generated fixtures are the pin.

Determinism protocol: top-k ties -> lowest index first (stable sort);
Gumbel noise is an explicit input.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Contraction precision.  The oracle proper is fp32 (the reference's inference dtype).  ``contractions("bf16")`` turns the
# SAME code into the torch-bf16 control of BASELINE.md section 4 / SURVEY.md 7(b,c): every nn.Linear on the block path and
# both attention matmuls take bf16 operands and return bf16 (torch accumulates in f32 and rounds the result once, like
# torch.autocast), everything else -- LayerNorm, softmax, RoPE, residual stream, scorers, token merge -- stays fp32.
# Runs on whatever device the inputs live on, so the control can use torch's own bf16 GEMMs on the same GPU as the HIP path.
# --------------------------------------------------------------------------------------
_CONTRACT = {"mode": "fp32"}


class contractions:
    def __init__(self, mode: str):
        assert mode in ("fp32", "bf16")
        self.mode = mode

    def __enter__(self):
        self.prev = _CONTRACT["mode"]
        _CONTRACT["mode"] = self.mode

    def __exit__(self, *a):
        _CONTRACT["mode"] = self.prev


def _linear(x, w, b=None):
    if _CONTRACT["mode"] == "bf16":
        return F.linear(x.bfloat16(), w.bfloat16(), None if b is None else b.bfloat16()).float()
    return F.linear(x, w, b)


def _matmul(a, b):
    if _CONTRACT["mode"] == "bf16":
        return (a.bfloat16() @ b.bfloat16()).float()
    return a @ b


LN_EPS = 1e-6            # toc3d_eva_vit.py:38 norm_layer=partial(nn.LayerNorm, eps=1e-6)
TORCH_LN_EPS = 1e-5      # nn.LayerNorm default, used inside the scorers (toc3d_utils.py:100,331; misc.py:172)
PAD_SCORE = -1e6         # toc3d_eva_vit.py:415


# --------------------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------------------
def window_partition(x: torch.Tensor, ws: int, pad_value: float = 0.0):
    """backbones/eva_utils.py:89-110.  x (B,H,W,C) -> (B*nW, ws, ws, C), (Hp, Wp)."""
    B, H, W, C = x.shape
    ph, pw = (-H) % ws, (-W) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph), value=pad_value)
    Hp, Wp = H + ph, W + pw
    x = x.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win: torch.Tensor, ws: int, pad_hw, hw):
    """backbones/eva_utils.py:113-133."""
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // ((Hp // ws) * (Wp // ws))
    x = win.reshape(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def abs_pos(pos_embed: torch.Tensor, has_cls: bool, hw):
    """backbones/eva_utils.py:229-258: drop cls row, bicubic resize (align_corners=False) to (h,w)."""
    h, w = hw
    p = pos_embed[:, 1:] if has_cls else pos_embed
    s = int(math.isqrt(p.shape[1]))
    assert s * s == p.shape[1]
    if s == h and s == w:
        return p.reshape(1, h, w, -1)
    g = F.interpolate(p.reshape(1, s, s, -1).permute(0, 3, 1, 2), size=(h, w), mode="bicubic", align_corners=False)
    return g.permute(0, 2, 3, 1)


def patch_embed(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor):
    """backbones/eva_utils.py:279-287: Conv2d(k=s=patch) then NCHW->NHWC."""
    p = w.shape[-1]
    if _CONTRACT["mode"] == "bf16":
        return F.conv2d(x.bfloat16(), w.bfloat16(), b.bfloat16(), stride=p).float().permute(0, 2, 3, 1)
    return F.conv2d(x, w, b, stride=p).permute(0, 2, 3, 1)


def layer_norm(x, w, b, eps=LN_EPS):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# --------------------------------------------------------------------------------------
# attention / MLP
# --------------------------------------------------------------------------------------
def rope_rotate(t: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor):
    """backbones/eva_utils.py:318-322,379: t*cos + rotate_half(t)*sin, pairs (2i,2i+1) -> (-x[2i+1], x[2i])."""
    t2 = t.reshape(*t.shape[:-1], -1, 2)
    rot = torch.stack((-t2[..., 1], t2[..., 0]), dim=-1).reshape(t.shape)
    return t * cos + rot * sin


def attention(x: torch.Tensor, sd, pre: str, num_heads: int, cos, sin):
    """backbones/eva_vit.py:86-119 and toc3d_eva_vit.py:484-518.

    x (B,N,C); cos/sin either (N,hd) shared by all windows (dense) or (B,N,hd) gathered per
    window by slot index (eva_utils.py:396-403).  Scale is applied after RoPE (eva_vit.py:110).
    """
    B, N, C = x.shape
    hd = C // num_heads
    q = _linear(x, sd[pre + "q_proj.weight"], sd[pre + "q_bias"])
    k = _linear(x, sd[pre + "k_proj.weight"], None)
    v = _linear(x, sd[pre + "v_proj.weight"], sd[pre + "v_bias"])
    q, k, v = (t.reshape(B, N, num_heads, hd).permute(0, 2, 1, 3) for t in (q, k, v))
    if cos.dim() == 3:
        cos, sin = cos[:, None], sin[:, None]
    q = rope_rotate(q, cos, sin)
    k = rope_rotate(k, cos, sin)
    q = q * hd ** -0.5
    a = _matmul(q, k.transpose(-2, -1)).softmax(dim=-1)
    o = _matmul(a, v).transpose(1, 2).reshape(B, N, C)
    return _linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def swiglu(x: torch.Tensor, sd, pre: str):
    """backbones/eva_vit.py:44-51 with subln=True (eva_vit.py:231): w3(LN(silu(w1 x) * (w2 x)))."""
    h = F.silu(_linear(x, sd[pre + "w1.weight"], sd[pre + "w1.bias"])) * _linear(x, sd[pre + "w2.weight"], sd[pre + "w2.bias"])
    h = layer_norm(h, sd[pre + "ffn_ln.weight"], sd[pre + "ffn_ln.bias"])
    return _linear(h, sd[pre + "w3.weight"], sd[pre + "w3.bias"])


def block_window_side(cfg, i: int) -> int:
    return cfg["global_window_size"] if i in cfg["global_attn_indexes"] else cfg["window_size"]


def dense_block(x: torch.Tensor, sd, i: int, cfg):
    """backbones/eva_vit.py:247-268.  LN *then* zero-pad: pad keys are k=0, v=v_bias."""
    pre = f"blocks.{i}."
    ws = block_window_side(cfg, i)
    H, W = x.shape[1:3]
    y = layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    y, pad_hw = window_partition(y, ws)
    nB = y.shape[0]
    y = attention(y.reshape(nB, ws * ws, -1), sd, pre + "attn.", cfg["num_heads"],
                  sd[pre + "attn.rope.freqs_cos"], sd[pre + "attn.rope.freqs_sin"])
    y = window_unpartition(y.reshape(nB, ws, ws, -1), ws, pad_hw, (H, W))
    x = x + y
    x = x + swiglu(layer_norm(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]), sd, pre + "mlp.")
    return x


# --------------------------------------------------------------------------------------
# token selection helpers
# --------------------------------------------------------------------------------------
def sort_desc_stable(score: torch.Tensor):
    """toc3d_utils.py:139 with the tie rule pinned: descending, equal scores keep ascending index."""
    return torch.sort(score, dim=1, descending=True, stable=True)


def gather_rows(x: torch.Tensor, idx: torch.Tensor):
    """toc3d_utils.py:28-44 batch_index_select for (B,N,C) or (B,N)."""
    if x.dim() == 3:
        return torch.gather(x, 1, idx[..., None].expand(-1, -1, x.shape[-1]))
    return torch.gather(x, 1, idx)


def merge_tokens(x_drop: torch.Tensor, s_drop: torch.Tensor):
    """toc3d_utils.py:65-70: sum_j (s_j / sum s) x_j  (weights from raw log-prob scores, pads are -1e6)."""
    w = s_drop / s_drop.sum(dim=1, keepdim=True)
    return (w[..., None] * x_drop).sum(dim=1, keepdim=True)


def accel_block(x: torch.Tensor, scores: torch.Tensor, ratio: float, sd, i: int, cfg,
                capture: Optional[dict] = None):
    """backbones/toc3d_eva_vit.py:395-477 (eval, use_represent_tokens=True).

    x (B,H,W,C); scores (B,H,W) image-level log-probs of the most recent scorer.
    """
    pre = f"blocks.{i}."
    ws = block_window_side(cfg, i)
    B, H, W, C = x.shape
    xw, pad_hw = window_partition(x, ws)                                      # :414  pad value 0, *before* LN
    sw, _ = window_partition(scores[..., None], ws, pad_value=PAD_SCORE)      # :415
    nW, N = xw.shape[0], ws * ws
    xw = xw.reshape(nW, N, C)
    sw = sw.reshape(nW, N)
    k = int(N * ratio)                                                        # toc3d_utils.py:138
    s_sorted, order = sort_desc_stable(sw)                                    # :419 -> toc3d_utils.py:139
    slow_idx, fast_idx = order[:, :k], order[:, k:]
    fast_score = s_sorted[:, k:]
    slow = gather_rows(xw, slow_idx)                                          # :421
    fast = gather_rows(xw, fast_idx)                                          # :424
    has_rep = fast.shape[1] > 0
    if has_rep:
        rep = merge_tokens(fast, fast_score)                                  # :427
        slow = torch.cat([slow, rep], dim=1)                                  # :430
        rope_idx = torch.cat([slow_idx, torch.full((nW, 1), k, dtype=torch.long, device=slow_idx.device)], dim=1)   # :434-435 (slot k)
    else:
        rope_idx = slow_idx
    cos = sd[pre + "attn.rope.freqs_cos"][rope_idx]                           # eva_utils.py:400-401
    sin = sd[pre + "attn.rope.freqs_sin"][rope_idx]
    # forward_slow, :366-386
    raw1 = attention(layer_norm(slow, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"]), sd, pre + "attn.",
                     cfg["num_heads"], cos, sin)
    slow = slow + raw1
    raw2 = swiglu(layer_norm(slow, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]), sd, pre + "mlp.")
    slow = slow + raw2
    if capture is not None:
        capture[f"block{i}.slow_idx"] = slow_idx
        capture[f"block{i}.slow_out"] = slow
    if has_rep:
        slow = slow[:, :-1]                                                   # :449
        fast = fast + raw1[:, -1:] + raw2[:, -1:]                             # :452-456
        out = torch.zeros_like(xw)                                            # :460
        out.scatter_(1, slow_idx[..., None].expand(-1, -1, C), slow)          # toc3d_utils.py:58
        out.scatter_(1, fast_idx[..., None].expand(-1, -1, C), fast)          # toc3d_utils.py:59
    else:
        out = slow                                                            # :463 (sorted order)
    return window_unpartition(out.reshape(nW, ws, ws, C), ws, pad_hw, (H, W))


# --------------------------------------------------------------------------------------
# scorers
# --------------------------------------------------------------------------------------
def pos2posemb(pos: torch.Tensor, num_feats: int, temperature: float = 10000.0):
    """utils/positional_encoding.py:14-37 core: per coordinate, interleaved sin/cos of 2*pi*p / T^(2*floor(i/2)/F)."""
    pos = pos * (2 * math.pi)
    dim_t = torch.arange(num_feats, dtype=torch.float32, device=pos.device)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_feats)
    a = pos[..., None] / dim_t
    return torch.stack((a[..., 0::2].sin(), a[..., 1::2].cos()), dim=-1).flatten(-2)


def pos2posemb3d(p: torch.Tensor):
    """utils/positional_encoding.py:14-26; concatenation order is (y, x, z) (:25)."""
    e = pos2posemb(p, 128)                      # (..., 3, 128)
    return torch.cat((e[..., 1, :], e[..., 0, :], e[..., 2, :]), dim=-1)


def pos2posemb1d(t: torch.Tensor):
    """utils/positional_encoding.py:28-37 (computed in t's dtype: float64 when timestamps are f64)."""
    return pos2posemb(t[..., 0], 256)


def nerf_encoding(x: torch.Tensor, n_freq: int = 6):
    """utils/positional_encoding.py:39-81 (include_input=False, log sampling): [sin(2^k x), cos(2^k x)] k-major."""
    parts = []
    for kk in range(n_freq):
        f = torch.tensor(2.0 ** kk, dtype=x.dtype, device=x.device)
        parts += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(parts, dim=-1)


def mln(x, c, sd, pre):
    """utils/misc.py:154-188: gamma(h) * LN_noaffine(x) + beta(h), h = relu(reduce(c))."""
    xn = F.layer_norm(x, (x.shape[-1],), None, None, TORCH_LN_EPS)
    h = F.relu(F.linear(c, sd[pre + "reduce.0.weight"], sd[pre + "reduce.0.bias"]))
    return F.linear(h, sd[pre + "gamma.weight"], sd[pre + "gamma.bias"]) * xn + F.linear(h, sd[pre + "beta.weight"], sd[pre + "beta.bias"])


def motion_aware_queries(sd, pre, queries, ref_points, vel, timestamp, ego_pose, ego_pose_inv):
    """backbones/toc3d_utils.py:334-360."""
    ones = torch.ones_like(ref_points[..., :1])
    pts = (ego_pose_inv[:, None] @ torch.cat([ref_points, ones], -1)[..., None])[..., :3, 0]     # misc.py:191-200
    pc = sd[pre + "pc_range"]
    pts = (pts - pc[:3]) / (pc[3:6] - pc[0:3])                                                   # :348
    pos = F.linear(F.relu(F.linear(pos2posemb3d(pts), sd[pre + "query_embedding.0.weight"], sd[pre + "query_embedding.0.bias"])),
                   sd[pre + "query_embedding.2.weight"], sd[pre + "query_embedding.2.bias"])      # :349
    motion = torch.cat([vel, timestamp, ego_pose[..., :3, :].flatten(-2)], dim=-1).float()        # :351 (f64 -> f32 cast)
    motion = nerf_encoding(motion)                                                                # :352
    pos = mln(pos, motion, sd, pre + "ego_pose_pe.")                                              # :353
    te = F.linear(pos2posemb1d(timestamp).float(), sd[pre + "time_embedding.0.weight"], sd[pre + "time_embedding.0.bias"])
    pos = pos + F.layer_norm(te, (te.shape[-1],), sd[pre + "time_embedding.1.weight"], sd[pre + "time_embedding.1.bias"], TORCH_LN_EPS)  # :354
    return mln(queries, motion, sd, pre + "ego_pose_queries.") + pos                               # :356-358


def query_based_score(x, mask, queries, sd, pre):
    """backbones/toc3d_utils.py:232-252 (score_type='attention', use_mask, attn_scale)."""
    B = x.shape[0]
    z = F.linear((x * mask).flatten(1, 2), sd[pre + "input_proj.0.weight"], sd[pre + "input_proj.0.bias"])
    q = queries.repeat_interleave(B // queries.shape[0], dim=0)
    att = torch.einsum("bnc,bqc->bnq", z, q) * (q.shape[-1] ** -0.5)
    return F.log_softmax(F.linear(att, sd[pre + "aggregate.0.weight"], sd[pre + "aggregate.0.bias"]), dim=-1)


def score_based_score(x, mask, sd, pre):
    """backbones/toc3d_utils.py:114-129 (first frame of a scene, prev_exists=False)."""
    B, H, W, C = x.shape
    t = (x * mask).reshape(B, H * W, C)
    t = F.gelu(F.linear(F.layer_norm(t, (C,), sd[pre + "in_conv.0.weight"], sd[pre + "in_conv.0.bias"], TORCH_LN_EPS),
                        sd[pre + "in_conv.1.weight"], sd[pre + "in_conv.1.bias"]))
    loc = t[:, :, : C // 2]
    glb = t[:, :, C // 2:].mean(dim=1, keepdim=True).expand(B, H * W, C // 2)
    t = torch.cat([loc, glb], dim=2)
    t = F.gelu(F.linear(t, sd[pre + "out_conv.0.weight"], sd[pre + "out_conv.0.bias"]))
    t = F.gelu(F.linear(t, sd[pre + "out_conv.2.weight"], sd[pre + "out_conv.2.bias"]))
    return F.log_softmax(F.linear(t, sd[pre + "out_conv.4.weight"], sd[pre + "out_conv.4.bias"]), dim=-1)


def sample_image_level(pred: torch.Tensor, ratio: float, g: torch.Tensor):
    """backbones/toc3d_utils.py:131-158 at image level: indices by stable descending sort of pred[...,0];
    new soft mask = softmax(pred + g)[...,0] (F.gumbel_softmax, tau=1, hard=False)."""
    score = pred[:, :, 0]
    k = int(score.shape[1] * ratio)
    _, order = sort_desc_stable(score)
    mask = torch.softmax(pred + g, dim=-1)[:, :, 0:1]
    return order[:, :k], order[:, k:], mask


# --------------------------------------------------------------------------------------
# whole backbones
# --------------------------------------------------------------------------------------
def stem(sd, cfg, img):
    x = patch_embed(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"])
    return x + abs_pos(sd["pos_embed"], cfg.get("pretrain_use_cls_token", True), (x.shape[1], x.shape[2]))


def forward_eva(sd, cfg, img, capture: Optional[dict] = None):
    """backbones/eva_vit.py:409-428 -> {'last_feat': (B,C,h,w)}."""
    x = stem(sd, cfg, img)
    for i in range(cfg["depth"]):
        x = dense_block(x, sd, i, cfg)
        if capture is not None:
            capture[f"block{i}.out"] = x
    return {"last_feat": x.permute(0, 3, 1, 2)}


def forward_toc3d(sd, cfg, img, temp_queries, temp_ref_points, temp_vel, temp_timestamp, temp_ego_pose,
                  ego_pose_inv, prev_exists: bool, gumbel: List[torch.Tensor], capture: Optional[dict] = None,
                  forced: Optional[List[tuple]] = None):
    """backbones/toc3d_eva_vit.py:230-310 (eval).  Returns dict(last_feat, token_masks, keep_idx, drop_idx).

    ``forced`` (tests only): per stage ``(score (B, H*W), mask (B, H*W))`` that REPLACE the scorer's image-level log-probs
    and soft mask, e.g. the real reference's own values from a golden fixture -- every later block then selects exactly the
    reference's tokens, which takes the top-k flips of a reduced-precision run out of the error."""
    x = stem(sd, cfg, img)
    B, H, W, C = x.shape
    if capture is not None:
        capture["stem"] = x
    masks = torch.ones(B, H, W, 1, device=x.device)                           # :251
    scores, ratio, stage = None, None, 0
    token_masks, keep_idx, drop_idx = [], [], []
    for i in range(cfg["depth"]):
        if i in cfg["pruning_loc"]:                                            # :264-285
            pre = f"score_predictor.{stage}."
            ratio = cfg["token_ratio"][stage]
            mq = motion_aware_queries(sd, pre, temp_queries, temp_ref_points, temp_vel, temp_timestamp,
                                      temp_ego_pose, ego_pose_inv)            # toc3d_utils.py:376-383 (always)
            if prev_exists:                                                    # toc3d_utils.py:270-273
                pred = query_based_score(x, masks, mq, sd, pre)
            else:
                pred = score_based_score(x, masks, sd, pre)
            ki, di, m = sample_image_level(pred, ratio, gumbel[stage])
            masks = m.reshape(B, H, W, 1)                                     # replaced, not multiplied (:266)
            scores = pred[:, :, 0].reshape(B, H, W)                           # toc3d_utils.py:415,420
            if forced is not None:
                scores = forced[stage][0].to(x.device).float().reshape(B, H, W)
                masks = forced[stage][1].to(x.device).float().reshape(B, H, W, 1)
                _, order = sort_desc_stable(scores.reshape(B, H * W))
                kk = int(H * W * ratio)
                ki, di = order[:, :kk], order[:, kk:]
            token_masks.append(masks)
            keep_idx.append(ki)
            drop_idx.append(di)
            if capture is not None:
                capture[f"stage{stage}.queries"] = mq
                capture[f"stage{stage}.pred"] = pred
            stage += 1
        accel = len(cfg["pruning_loc"]) > 0 and i >= cfg["pruning_loc"][0] and (
            cfg.get("accelerate_global", True) or i not in cfg["global_attn_indexes"])   # :178-180
        if accel:
            x = accel_block(x, scores, ratio, sd, i, cfg, capture)
        else:
            x = dense_block(x, sd, i, cfg)
        if capture is not None:
            capture[f"block{i}.out"] = x
    return {"last_feat": x.permute(0, 3, 1, 2), "token_masks": token_masks, "keep_idx": keep_idx, "drop_idx": drop_idx}


def cpfpn(sd, feat: torch.Tensor):
    """necks/cp_fpn.py:156-208 for in_channels=[C], num_outs=2: 1x1 lateral, 3x3 fpn conv, stride-2 subsample."""
    lat = F.conv2d(feat, sd["lateral_convs.0.conv.weight"], sd["lateral_convs.0.conv.bias"])
    o0 = F.conv2d(lat, sd["fpn_convs.0.conv.weight"], sd["fpn_convs.0.conv.bias"], padding=1)
    return o0, F.max_pool2d(o0, 1, stride=2)
