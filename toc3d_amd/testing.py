"""Test instruments for the backbones -- NOT part of the product forward.

``instrument(model)`` turns a built ``ToC3DEVAViT`` into :class:`InstrumentedToC3DEVAViT` in place (same parameters, same buffers).  The
instrumented model launches every frame eagerly and adds what the parity tests need:

* ``forced_scores=`` on ``forward``: per scorer stage ``(score (B*Nv, T), mask (B*Nv, T))`` that REPLACE the stage's image-level log-probs and
  soft mask -- e.g. the reference's own, from a golden fixture -- so that every block selects exactly the reference's tokens and only
  arithmetic differs (bf16 parity with forced selection);
* ``block_hook``: ``callable(i, group_plan, carried)`` after block ``i`` of a view group (per-block error budgets).
"""
from __future__ import annotations

import torch

from .backbone import ToC3DEVAViT


class InstrumentedToC3DEVAViT(ToC3DEVAViT):
    _instrumented = True
    block_hook = None
    _forced = None

    @torch.no_grad()
    def forward(self, x, *args, forced_scores=None, **kwargs):
        self._forced = None
        if forced_scores is not None:
            dev = x.device
            self._forced = [(f[0].to(device=dev, dtype=torch.float32).reshape(-1), f[1].to(device=dev, dtype=torch.float32).reshape(-1)) for f in forced_scores]
        try:
            return super().forward(x, *args, **kwargs)
        finally:
            self._forced = None

    def _stage_override(self, st, plan, score, mask):
        if self._forced is not None:
            T = plan["T"]
            r0, r1 = plan["v0"] * T, (plan["v0"] + plan["nv"]) * T
            score.copy_(self._forced[st][0][r0:r1])
            mask.copy_(self._forced[st][1][r0:r1])

    def _block_done(self, i, plan, carried):
        if self.block_hook is not None:
            self.block_hook(i, plan, carried)


def instrument(model: ToC3DEVAViT) -> InstrumentedToC3DEVAViT:
    assert isinstance(model, ToC3DEVAViT)
    model.__class__ = InstrumentedToC3DEVAViT
    return model
