"""StreamPETR temporal memory bank on the GPU -- the head-side producer of the backbone scorer's inputs
(SURVEY.md section 8f row 3).

Mirrors the memory part of ``StreamPETRHead`` (``dense_heads/streampetr_head.py``): ``reset_memory`` :315-320,
``pre_update_memory`` :322-346, ``post_update_memory`` :348-377, with the same attribute names
(``memory_embedding``, ``memory_reference_point``, ``memory_timestamp``, ``memory_egopose``, ``memory_velo``), plus
``backbone_queries`` = the slice ``Petr3D.extract_img_feat`` hands to the backbone (``detectors/petr3d.py:115-134``).
All updates run as HIP kernels on the caller's stream (``toc3d_memory_*``); there is no CPU path.

Differences to the reference, by design: the bank lives in two fixed-capacity (memory_len + topk) buffers that swap roles
instead of being re-allocated by ``torch.cat`` every frame; ``memory_timestamp`` is float64 from the start (the reference's
is float32 zeros until the first ``post_update_memory`` promotes it, :371); ``torch.topk`` ties resolve to the lowest index.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import lib


class TemporalMemory:
    def __init__(self, memory_len: int = 512, topk_proposals: int = 128, num_propagated: int = 128, embed_dims: int = 256,
                 pc_range: Sequence[float] = (-51.2, -51.2, -5.0, 51.2, 51.2, 3.0), pseudo_reference_points: Optional[torch.Tensor] = None,
                 device="cuda"):
        self.memory_len, self.topk_proposals, self.num_propagated, self.embed_dims = memory_len, topk_proposals, num_propagated, embed_dims
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("toc3d_amd.TemporalMemory runs on an AMD GPU through libtoc3d_gfx950.so; there is no CPU path")
        lib.load()
        self.pc_range = torch.tensor(list(pc_range), dtype=torch.float32, device=self.device)
        if num_propagated > 0:
            if pseudo_reference_points is None:       # nn.init.uniform_(pseudo_reference_points.weight, 0, 1), :303-306
                pseudo_reference_points = torch.rand(num_propagated, 3)
            assert tuple(pseudo_reference_points.shape) == (num_propagated, 3)
            self.pseudo_reference_points = pseudo_reference_points.detach().float().contiguous().to(self.device)
        else:
            self.pseudo_reference_points = None
        self._banks = None
        self.reset_memory()

    # -- state ------------------------------------------------------------------------------------------
    def reset_memory(self):                                                   # :315-320
        self._cur, self._len, self._B = 0, 0, 0

    def _alloc(self, B):
        cap, D, dev = self.memory_len + self.topk_proposals, self.embed_dims, self.device
        mk = lambda *s, dt=torch.float32: torch.zeros(B, cap, *s, dtype=dt, device=dev)
        self._banks = [dict(emb=mk(D), ref=mk(3), ts=mk(dt=torch.float64), pose=mk(4, 4), vel=mk(2)) for _ in range(2)]
        self._score = torch.empty(0, device=dev)
        self._B = B

    def _view(self, key):
        return None if self._len == 0 else self._banks[self._cur][key][:, :self._len]

    memory_embedding = property(lambda self: self._view("emb"))
    memory_reference_point = property(lambda self: self._view("ref"))
    memory_egopose = property(lambda self: self._view("pose"))
    memory_velo = property(lambda self: self._view("vel"))

    @property
    def memory_timestamp(self):
        t = self._view("ts")
        return None if t is None else t.unsqueeze(-1)

    @staticmethod
    def _dev(t, dtype, dev):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError("toc3d_amd.TemporalMemory: inputs must be CUDA/HIP tensors (no CPU path)")
        return t.to(device=dev, dtype=dtype).contiguous()

    # -- updates ----------------------------------------------------------------------------------------
    def pre_update_memory(self, data: Dict[str, torch.Tensor]):
        x = self._dev(data["prev_exists"], torch.float32, self.device).flatten()
        B = x.shape[0]
        fresh = self._len == 0
        if fresh or B != self._B:
            self._alloc(B)
            fresh = True
        elif self._len != self.memory_len + self.topk_proposals:
            raise RuntimeError("pre_update_memory called twice without post_update_memory")
        bank = self._banks[self._cur]
        if fresh:
            for t in bank.values():
                t.zero_()
        ts = self._dev(data["timestamp"], torch.float64, self.device).flatten()
        inv = self._dev(data["ego_pose_inv"], torch.float32, self.device).reshape(B, 4, 4)
        lib.call("toc3d_memory_pre_update", bank["emb"], bank["ref"], bank["ts"], bank["pose"], bank["vel"], x, ts, inv,
                 self.pseudo_reference_points, self.pc_range, B, self.memory_len + self.topk_proposals, self.memory_len,
                 self.num_propagated, self.embed_dims, int(fresh), lib.stream_ptr())
        self._len = self.memory_len

    def post_update_memory(self, data, rec_ego_pose, all_cls_scores, all_bbox_preds, outs_dec, mask_dict=None):
        """Inference form of :348-377 (``mask_dict`` is the training-time denoising split; not built)."""
        if mask_dict:
            raise NotImplementedError("training-time denoising queries (mask_dict) are out of scope")
        if self._len != self.memory_len:
            raise RuntimeError("post_update_memory needs a preceding pre_update_memory")
        dev = self.device
        cls = self._dev(all_cls_scores[-1], torch.float32, dev)
        bbox = self._dev(all_bbox_preds[-1], torch.float32, dev)
        dec = self._dev(outs_dec[-1], torch.float32, dev)
        pose = self._dev(rec_ego_pose, torch.float32, dev)
        B, Q, ncls = cls.shape
        assert B == self._B and bbox.shape[:2] == (B, Q) and dec.shape == (B, Q, self.embed_dims) and pose.shape == (B, Q, 4, 4)
        s = lib.stream_ptr()
        if self._score.numel() != B * Q:
            self._score = torch.empty(B, Q, dtype=torch.float32, device=dev)
            self._order = torch.empty(B, Q, dtype=torch.int64, device=dev)
        lib.call("toc3d_memory_scores", cls, B * Q, ncls, self._score, s)
        lib.call("toc3d_rank_desc", self._score, B, Q, self._order, s)
        src, dst = self._banks[self._cur], self._banks[1 - self._cur]
        ego = self._dev(data["ego_pose"], torch.float32, dev).reshape(B, 4, 4)
        ts = self._dev(data["timestamp"], torch.float64, dev).flatten()
        lib.call("toc3d_memory_post_update", src["emb"], src["ref"], src["ts"], src["pose"], src["vel"], dst["emb"], dst["ref"], dst["ts"],
                 dst["pose"], dst["vel"], self._order, pose, bbox, bbox.shape[2], dec, ego, ts, B, Q, self.memory_len + self.topk_proposals,
                 self.memory_len, self.topk_proposals, self.embed_dims, s)
        self._cur = 1 - self._cur
        self._len = self.memory_len + self.topk_proposals

    # -- consumer side: what Petr3D passes to the backbone (detectors/petr3d.py:115-134) --------------------
    def backbone_queries(self, num_proposals: int, prev_exists, batch_size: int = 1) -> Dict[str, torch.Tensor]:
        mid = bool(prev_exists.bool().flatten()[0].item()) if isinstance(prev_exists, torch.Tensor) else bool(prev_exists)
        if not mid or self._len == 0:
            B = self._B or batch_size
            z = lambda *s, dt=torch.float32: torch.zeros(B, num_proposals, *s, dtype=dt, device=self.device)
            return dict(temp_queries=z(self.embed_dims), temp_ref_points=z(3), temp_timestamp=z(1), temp_ego_pose=z(4, 4), temp_vel=z(2),
                        prev_exists=False)
        n = num_proposals
        return dict(temp_queries=self.memory_embedding[:, :n], temp_ref_points=self.memory_reference_point[:, :n],
                    temp_timestamp=self.memory_timestamp[:, :n], temp_ego_pose=self.memory_egopose[:, :n], temp_vel=self.memory_velo[:, :n],
                    prev_exists=True)
