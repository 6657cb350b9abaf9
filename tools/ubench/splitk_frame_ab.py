#!/usr/bin/env python3
"""Development A/B (round 5): deterministic split-K picks for the N = 1024 residual GEMMs against the shipped table BY FRAME TIME -- replayed launch plans,
alternating timed regions in one process (cdna_hip_programming.md rule 24).
    python tools/ubench/splitk_frame_ab.py [rounds]
Legs: shipped table | the split-K variant that came closest cold on the small w3 launches | split 2 of tile 16 on every N = 1024 launch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import toc3d_amd
from toc3d_amd import configs, lib, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
name, H, W, dev = "toc3d_faster", 320, 800, "cuda:0"
cfg = configs.get(name)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shipped = os.path.join(root, "toc3d_amd", "tuned", f"{name}_{H}x{W}_bf16.json")
m = toc3d_amd.build_backbone(dict(cfg, precision="bf16"))
m.load_state_dict(synth.make_state_dict(cfg))
m = m.to(dev).eval()
m.alias_outputs = True
m.load_tuning(shipped)
neck = toc3d_amd.build_neck(dict(configs.CPFPN_CFG, precision="bf16"))
neck.load_state_dict(synth.neck_state_dict(configs.CPFPN_CFG))
neck = neck.to(dev).eval()
neck.alias_outputs = True
neck._tuned = m._tuned
inp = synth.make_inputs(cfg, views_per_frame=6, hw=(H, W))
x = inp["x"].to(dev)
kw = {k: inp[k].to(dev) for k in ("temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")}
g = [t.to(dev) for t in inp["gumbel"]]


def step():
    return neck([m(x, prev_exists=True, gumbel_noise=g, **kw).img_feats["last_feat"]])


def forget_plans():
    for p in (m._plans or {}).values():
        p.pop("launch", None)
    for ws in neck._ws.values():
        ws.pop("launch", None)


def measure(tab, frames=40):
    m._tuned.clear()
    m._tuned.update(tab)
    forget_plans()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames


step()
torch.cuda.synchronize()
base = dict(m._tuned)
n1024 = [k for k in base if k[0] in (5, 6) and k[2] == 1024 and k[1] <= 6000]
legs = {"shipped table": base,
        "w3 at M <= 2898 as split 2 of the 64x128 tile (v2010: 1.07-1.2x of the best unsplit cold)": {**base, **{k: 2010 for k in n1024 if k[0] == 5 and k[1] <= 2898}},
        "proj at M <= 2898 as split 2 (v2026)": {**base, **{k: 2026 for k in n1024 if k[0] == 6 and k[1] <= 2898}},
        "every N = 1024 launch as split 2 of tile 16 (v2016)": {**base, **{k: 2016 for k in n1024}}}
times = {k: [] for k in legs}
for _ in range(rounds):
    for k, tab in legs.items():
        times[k].append(measure(tab))
med = lambda ts: sorted(ts)[len(ts) // 2]
print(f"# frame time ms (backbone + neck, replayed launch plans), {rounds} alternations of 40 frames; {len(n1024)} N = 1024 table entries")
for k, ts in times.items():
    print(f"{k:100s}: " + " ".join(f"{1e3 * t:.4f}" for t in ts) + f"   median {1e3 * med(ts):.4f} = {1 / med(ts):.1f} frames/s", flush=True)
