"""Generates tests/golden/head_tokens.npz from the REAL reference: StreamPETRHead.position_embeding (unbound) and the reference's
MLN / SELayer_Linear modules, composed as StreamPETRHead.forward :627-639 does.  Runs only in the build container."""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as R              # noqa: E402
from oracle import head_tokens_oracle as HO      # noqa: E402
from toc3d_amd import synth                      # noqa: E402

CFG, B, N, H, W = synth.HEAD_TOKENS_TINY, 1, 2, 4, 6


def main():
    ref = R.load_reference()
    head = R.load_reference_head()
    sd = synth.head_tokens_state_dict(CFG)
    inp = synth.head_tokens_inputs(CFG, B, N, H, W)
    pad_h, pad_w = H * CFG["stride"], W * CFG["stride"]
    E, C = CFG["embed_dims"], CFG["in_channels"]
    obj = types.SimpleNamespace()
    obj.coords_d = HO.coords_d(CFG["position_range"], CFG["depth_num"], CFG["depth_start"], CFG["LID"])
    obj.position_range = torch.tensor(CFG["position_range"])
    obj.position_encoder = nn.Sequential(nn.Linear(CFG["depth_num"] * 3, 4 * E), nn.ReLU(), nn.Linear(4 * E, E))
    memory_embed = nn.Sequential(nn.Linear(C, E), nn.ReLU(), nn.Linear(E, E))
    mln = ref.misc.MLN(8, f_dim=E)
    se = ref.misc.SELayer_Linear(E)
    for mod, pre in ((obj.position_encoder, "position_encoder."), (memory_embed, "memory_embed."), (mln, "spatial_alignment."), (se, "featurized_pe.")):
        mod.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
    # mmdet's inverse_sigmoid is stubbed as None in the harness: give the head module the published definition
    sys.modules[head.__module__].inverse_sigmoid = HO.inverse_sigmoid
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and a[0] in ("cuda", "cpu")) else orig_to(self, *a, **k)   # :404 hops cpu -> cuda
    try:
        with torch.no_grad():
            centers = ref.misc.locations(inp["feats"].flatten(0, 1), CFG["stride"], pad_h, pad_w)[None].repeat(B * N, 1, 1, 1)
            data = dict(intrinsics=inp["intrinsics"], lidar2img=inp["lidar2img"])
            pos, cone = head.position_embeding(obj, data, centers, None, [dict(pad_shape=[(pad_h, pad_w, 3)])])
            memory = inp["feats"].permute(0, 1, 3, 4, 2).reshape(B, N * H * W, C)
            memory = mln(memory_embed(memory), cone)
            pos_out = se(pos, memory)
    finally:
        torch.Tensor.to = orig_to
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "head_tokens.npz")
    np.savez_compressed(path, pos_raw=pos.numpy(), cone=cone.numpy(), memory=memory.numpy(), pos_embed=pos_out.numpy())
    print("wrote", path, pos.shape, cone.shape, memory.shape)


if __name__ == "__main__":
    main()
