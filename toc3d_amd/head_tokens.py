"""Token side of ``StreamPETRHead.forward`` on the GPU (second half): the consumers of the neck's
features right behind the backbone -- ``position_embeding`` (``dense_heads/streampetr_head.py:378-422``), ``memory_embed``,
``spatial_alignment`` (``MLN(8)``) and ``featurized_pe`` (``SELayer_Linear``), composed as ``forward`` :627-639 does.

Sub-module names equal the head's, so the ``pts_bbox_head.*`` keys of a reference checkpoint load unchanged
(``position_encoder.{0,2}``, ``memory_embed.{0,2}``, ``spatial_alignment.{reduce.0,gamma,beta}``, ``featurized_pe.{conv_reduce,
conv_expand}``).  Linear layers run on ``toc3d_linear``; geometry / LayerNorm / gates on ``toc3d_head_*`` kernels.  No CPU path.
``torch.linalg.inv`` of the (B*N) 4x4 ``lidar2img`` matrices is plumbing on the device (the reference hops to the CPU for it, :404).
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn

from . import lib


def _ru(x, m):
    return (x + m - 1) // m * m


class _MLN(nn.Module):                      # parameter container: models/utils/misc.py:161-172
    def __init__(self, c_dim, f_dim):
        super().__init__()
        self.reduce = nn.Sequential(nn.Linear(c_dim, f_dim), nn.ReLU())
        self.gamma, self.beta = nn.Linear(f_dim, f_dim), nn.Linear(f_dim, f_dim)


class _SE(nn.Module):                       # models/utils/misc.py:140-145
    def __init__(self, ch):
        super().__init__()
        self.conv_reduce, self.conv_expand = nn.Linear(ch, ch), nn.Linear(ch, ch)


class HeadTokenEmbedding(nn.Module):
    def __init__(self, in_channels=256, embed_dims=256, depth_num=64, depth_start=1.0, LID=True, stride=16,
                 position_range: Sequence[float] = (-61.2, -61.2, -10.0, 61.2, 61.2, 10.0), precision="fp32", **unused):
        super().__init__()
        assert precision in ("bf16", "fp32") and embed_dims <= 1024 and depth_num >= 30
        self.in_channels, self.embed_dims, self.depth_num, self.stride, self.precision = in_channels, embed_dims, depth_num, stride, precision
        E = embed_dims
        self.position_encoder = nn.Sequential(nn.Linear(depth_num * 3, 4 * E), nn.ReLU(), nn.Linear(4 * E, E))       # :262-266
        self.memory_embed = nn.Sequential(nn.Linear(in_channels, E), nn.ReLU(), nn.Linear(E, E))                      # :268-272
        self.spatial_alignment = _MLN(8, E)                                                                           # :288
        self.featurized_pe = _SE(E)                                                                                   # :275
        pr = torch.tensor(list(position_range), dtype=torch.float32)
        index = torch.arange(0, depth_num, 1).float()                                                                 # :221-232
        if LID:
            cd = depth_start + (pr[3] - depth_start) / (depth_num * (1 + depth_num)) * index * (index + 1)
        else:
            cd = depth_start + (pr[3] - depth_start) / depth_num * index
        self.register_buffer("coords_d", cd, persistent=False)
        self._pr = pr                                                       # host copy: the C ABI takes position_range from the host
        self._packed = None
        self._ws = {}

    def _load_from_state_dict(self, *a, **k):
        self._packed = None
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def _pack(self, dev):
        dt = lib.BF16 if self.precision == "bf16" else lib.F32
        tdt = torch.bfloat16 if self.precision == "bf16" else torch.float32
        s = lib.stream_ptr()

        def pack(lin):
            wgt = lin.weight.detach().float().contiguous()
            N, K = wgt.shape
            out = torch.empty(_ru(N, 128), _ru(K, 64), dtype=tdt, device=dev)
            lib.call("toc3d_pack_weight", dt, wgt, N, K, out, out.shape[0], out.shape[1], s)
            return out, lin.bias.detach().float().contiguous()
        P = dict(dt=dt, tdt=tdt, pe0=pack(self.position_encoder[0]), pe2=pack(self.position_encoder[2]), me0=pack(self.memory_embed[0]),
                 me2=pack(self.memory_embed[2]), red=pack(self.spatial_alignment.reduce[0]), gam=pack(self.spatial_alignment.gamma),
                 bet=pack(self.spatial_alignment.beta), se1=pack(self.featurized_pe.conv_reduce), se2=pack(self.featurized_pe.conv_expand))
        torch.cuda.current_stream().synchronize()
        return P

    @torch.no_grad()
    def forward(self, img_feats: torch.Tensor, intrinsics: torch.Tensor, lidar2img: torch.Tensor, pad_shape):
        """img_feats (B, N, C, h, w) f32 (neck level 0, ``data['img_feats']`` :626); intrinsics / lidar2img (B, N, 4, 4);
        pad_shape = (pad_h, pad_w[, 3]) of ``img_metas[0]['pad_shape'][0]``.  Returns (memory, pos_embed, cone): memory and pos_embed
        f32 (B, N*h*w, embed_dims), cone f32 (B, N*h*w, 8) (:419-421) -- all three freshly allocated, none aliases a workspace."""
        if not isinstance(img_feats, torch.Tensor) or not img_feats.is_cuda:
            raise RuntimeError("toc3d_amd.HeadTokenEmbedding: inputs must be CUDA/HIP tensors -- the HIP extension is the only compute path")
        dev = img_feats.device
        if self._packed is None:
            self._packed = self._pack(dev)
        P = self._packed
        dt, tdt = P["dt"], P["tdt"]
        B, N, C, h, w = img_feats.shape
        assert C == self.in_channels
        E, D = self.embed_dims, self.depth_num
        M, s = B * N * h * w, lib.stream_ptr()
        key = (B, N, h, w)
        if key not in self._ws:
            z = lambda r, c, d=tdt: torch.zeros(r, c, dtype=d, device=dev)
            self._ws[key] = dict(pin=z(M, _ru(3 * D, 64)), cone_a=z(M, 64), cone=z(M, 8, torch.float32), h1=z(M, _ru(4 * E, 64)), feat=z(M, _ru(C, 64)),
                                 m1=z(M, _ru(E, 64)), c1=z(M, _ru(E, 64)), mem_a=z(M, _ru(E, 64)), s1=z(M, _ru(E, 64)),
                                 pos=z(M, E, torch.float32), mem_raw=z(M, E, torch.float32), gam=z(M, E, torch.float32), bet=z(M, E, torch.float32),
                                 se=z(M, E, torch.float32))
        W = self._ws[key]
        memory = torch.empty(M, E, dtype=torch.float32, device=dev)
        pos_embed = torch.empty(M, E, dtype=torch.float32, device=dev)
        img2lidar = torch.linalg.inv(lidar2img.to(dev).float().reshape(B * N, 4, 4)).contiguous()
        intr = intrinsics.to(dev).float().reshape(B * N, 4, 4).contiguous()

        def linear(x, wb, out, n, k, f32_out=False, relu=False):
            wgt, b = wb
            epi = lib.EPI_RESIDUAL if f32_out else lib.EPI_BIAS
            lib.call("toc3d_linear", dt, epi, x, x.shape[1], wgt, wgt.shape[1], b, out, out.shape[1], None, 0, 0, None, None, M, n, wgt.shape[1], 0, s)
            if relu:
                lib.call("toc3d_relu_inplace", dt, out, out.numel(), s)

        # position_embeding :378-416
        lib.call("toc3d_head_frustum_inputs", dt, img2lidar, intr, self.coords_d, self._pr, B, N, h, w, D, self.stride, int(pad_shape[0]),
                 int(pad_shape[1]), W["pin"], W["pin"].shape[1], W["cone_a"], 64, W["cone"], s)
        linear(W["pin"], P["pe0"], W["h1"], 4 * E, 3 * D, relu=True)
        linear(W["h1"], P["pe2"], W["pos"], E, 4 * E, f32_out=True)
        # memory_embed :635
        lib.call("toc3d_nchw_to_rows", dt, img_feats.float().contiguous(), W["feat"], W["feat"].shape[1], B * N, C, h * w, s)
        linear(W["feat"], P["me0"], W["m1"], E, C, relu=True)
        linear(W["m1"], P["me2"], W["mem_raw"], E, E, f32_out=True)
        # spatial_alignment :638
        linear(W["cone_a"], P["red"], W["c1"], E, 8, relu=True)
        linear(W["c1"], P["gam"], W["gam"], E, E, f32_out=True)
        linear(W["c1"], P["bet"], W["bet"], E, E, f32_out=True)
        lib.call("toc3d_mln_apply", dt, W["mem_raw"], W["gam"], W["bet"], M, E, memory, W["mem_a"], W["mem_a"].shape[1], s)
        # featurized_pe :639
        linear(W["mem_a"], P["se1"], W["s1"], E, E, relu=True)
        linear(W["s1"], P["se2"], W["se"], E, E, f32_out=True)
        lib.call("toc3d_se_gate", W["pos"], W["se"], pos_embed, M * E, s)
        return memory.view(B, N * h * w, E), pos_embed.view(B, N * h * w, E), W["cone"].clone().view(B, N * h * w, 8)
