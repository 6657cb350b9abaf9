"""TEST INFRASTRUCTURE ONLY -- CPU (eager torch) restatement of StreamPETR's temporal memory bank: the producer of the
backbone scorer's inputs (``temp_queries`` ... ``temp_ego_pose``).

Follows ``dense_heads/streampetr_head.py``: ``reset_memory`` :315-320, ``pre_update_memory`` :322-346,
``post_update_memory`` :348-377, and the slice the detector hands to the backbone, ``detectors/petr3d.py:115-134``.
Helpers: ``models/utils/misc.py`` ``memory_refresh`` :7-11, ``topk_gather`` :13-23, ``transform_reference_points``.

Pinned: ``oracle/gen_golden.py`` drives the reference's own methods (``ref_harness.ReferenceMemory``) over a 4-frame
sequence (scene start, two continuation frames, scene change) and commits ``tests/golden/memory_bank.npz``.
Tie rule of ``torch.topk`` pinned to lowest index first, as for ``torch.sort`` elsewhere.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


def transform_reference_points(pts: torch.Tensor, pose: torch.Tensor) -> torch.Tensor:
    """misc.py ``transform_reference_points(reverse=False)``: homogeneous points (B, L, 3) through pose (B, 4, 4)."""
    hom = torch.cat([pts, torch.ones_like(pts[..., :1])], dim=-1)
    return (pose.unsqueeze(1) @ hom.unsqueeze(-1)).squeeze(-1)[..., :3]


class MemoryBank:
    """State = the five ``memory_*`` tensors of the head."""

    def __init__(self, memory_len: int, topk_proposals: int, num_propagated: int, embed_dims: int, pc_range, pseudo_reference_points: torch.Tensor):
        self.memory_len, self.topk, self.num_propagated, self.embed_dims = memory_len, topk_proposals, num_propagated, embed_dims
        self.pc_range = torch.tensor(pc_range, dtype=torch.float32)
        self.pseudo = pseudo_reference_points.float()
        self.reset_memory()

    def reset_memory(self):                                                        # :315-320
        self.embedding = self.reference_point = self.timestamp = self.egopose = self.velo = None

    def pre_update_memory(self, data: Dict[str, torch.Tensor]):                    # :322-346
        x = data["prev_exists"]
        B, L = x.shape[0], self.memory_len
        if self.embedding is None:
            self.embedding = x.new_zeros(B, L, self.embed_dims)
            self.reference_point = x.new_zeros(B, L, 3)
            self.timestamp = x.new_zeros(B, L, 1)
            self.egopose = x.new_zeros(B, L, 4, 4)
            self.velo = x.new_zeros(B, L, 2)
        else:
            ts = self.timestamp + data["timestamp"].unsqueeze(-1).unsqueeze(-1).to(self.timestamp.dtype)     # in-place += keeps the dtype
            pose = data["ego_pose_inv"].unsqueeze(1) @ self.egopose
            ref = transform_reference_points(self.reference_point, data["ego_pose_inv"])
            k = lambda t: t[:, :L] * x.view(-1, *([1] * (t.dim() - 1)))          # memory_refresh
            self.timestamp, self.reference_point, self.embedding = k(ts), k(ref), k(self.embedding)
            self.egopose, self.velo = k(pose), k(self.velo)
        if self.num_propagated > 0:
            P = self.num_propagated
            pseudo = self.pseudo * (self.pc_range[3:6] - self.pc_range[0:3]) + self.pc_range[0:3]
            self.reference_point = self.reference_point.clone()
            self.egopose = self.egopose.clone()
            self.reference_point[:, :P] = self.reference_point[:, :P] + (1 - x).view(B, 1, 1) * pseudo
            self.egopose[:, :P] = self.egopose[:, :P] + (1 - x).view(B, 1, 1, 1) * torch.eye(4)

    def post_update_memory(self, data, rec_ego_pose, cls_scores, bbox_preds, outs_dec):
        """Last decoder layer's outputs: cls_scores (B, Q, ncls), bbox_preds (B, Q, 10), outs_dec (B, Q, D); :355-377."""
        rec_ref, rec_velo, rec_mem = bbox_preds[..., :3], bbox_preds[..., -2:], outs_dec
        score = cls_scores.sigmoid().max(dim=-1, keepdim=True).values            # .topk(1, dim=-1).values[..., 0:1]
        idx = torch.sort(score, dim=1, descending=True, stable=True).indices[:, :self.topk]       # pinned tie rule
        g = lambda t: torch.gather(t, 1, idx.view(idx.shape[0], self.topk, *([1] * (t.dim() - 2))).expand(-1, -1, *t.shape[2:]))
        rec_ts = torch.zeros_like(score, dtype=torch.float64)
        self.embedding = torch.cat([g(rec_mem), self.embedding], dim=1)
        self.timestamp = torch.cat([g(rec_ts), self.timestamp], dim=1)              # promotes the bank to f64 (:371)
        self.egopose = torch.cat([g(rec_ego_pose), self.egopose], dim=1)
        self.reference_point = torch.cat([g(rec_ref), self.reference_point], dim=1)
        self.velo = torch.cat([g(rec_velo), self.velo], dim=1)
        self.reference_point = transform_reference_points(self.reference_point, data["ego_pose"])
        self.timestamp = self.timestamp - data["timestamp"].unsqueeze(-1).unsqueeze(-1)
        self.egopose = data["ego_pose"].unsqueeze(1) @ self.egopose

    def backbone_queries(self, num_proposals: int, mid_frame: bool) -> Dict[str, Optional[torch.Tensor]]:
        """detectors/petr3d.py:115-134."""
        if not mid_frame or self.embedding is None:
            B = 1 if self.embedding is None else self.embedding.shape[0]
            z = lambda *s: torch.zeros(B, num_proposals, *s)
            return dict(temp_queries=z(self.embed_dims), temp_ref_points=z(3), temp_timestamp=z(1), temp_ego_pose=z(4, 4), temp_vel=z(2))
        n = num_proposals
        return dict(temp_queries=self.embedding[:, :n], temp_ref_points=self.reference_point[:, :n], temp_timestamp=self.timestamp[:, :n],
                    temp_ego_pose=self.egopose[:, :n], temp_vel=self.velo[:, :n])

    def state(self):
        return dict(embedding=self.embedding, reference_point=self.reference_point, timestamp=self.timestamp, egopose=self.egopose, velo=self.velo)
