#!/usr/bin/env python3
"""Soak test (development aid): N ToC3D_faster bf16 forwards with two view groups, every output compared bit for bit with the first."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import toc3d_amd
from toc3d_amd import configs, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = configs.get("toc3d_faster"); sd = synth.make_state_dict(cfg)
inp = synth.make_inputs(cfg, views_per_frame=6)
d = lambda t: t.cuda()
kw = dict(temp_queries=d(inp["temp_queries"]), prev_exists=True, temp_ref_points=d(inp["temp_ref_points"]), temp_vel=d(inp["temp_vel"]),
          temp_timestamp=d(inp["temp_timestamp"]), temp_ego_pose=d(inp["temp_ego_pose"]), ego_pose_inv=d(inp["ego_pose_inv"]), gumbel_noise=[g.cuda() for g in inp["gumbel"]])
x = d(inp["x"])
bad = 0
for groups in (2, 1):
    m = toc3d_amd.build_backbone(dict(cfg, precision="bf16")); m.load_state_dict(sd); m = m.cuda().eval(); m.view_groups = groups
    ref = m(x, **kw).img_feats["last_feat"].clone()
    for i in range(n):
        if not torch.equal(m(x, **kw).img_feats["last_feat"], ref):
            bad += 1
    print(f"groups={groups}: {bad} of {n} forwards differ", flush=True)
sys.exit(1 if bad else 0)
