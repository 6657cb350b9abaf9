#!/usr/bin/env python3
"""Development measurement (round 5): ONE frame as two concurrent half-frames.  Every launch of the frame carries ~14 us of ramp + tail + epilogue around a K loop that is
throughput-bound (profiles/r05_two_tile_heights.txt); two independent chains on two HIP streams (views 0-2 | views 3-5, shared CUs -- round 3's split used disjoint CU
masks and lost) could fill each other's ramps and tails.  Same weights, same tile table (autotune fills the half-size shapes), replayed plans.
    python tools/ubench/two_half_frames.py [precision] [rounds] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import toc3d_amd
from toc3d_amd import configs, synth

precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda", 0)
name, H, W = "toc3d_faster", 320, 800
cfg = configs.get(name)
sd = synth.make_state_dict(cfg)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
table = os.path.join(root, "toc3d_amd", "tuned", f"{name}_{H}x{W}_{precision}.json")


def make(views):
    inp = synth.make_inputs(cfg, n_frames=1, views_per_frame=views, hw=(H, W))
    x = inp["x"].to(dev)
    kw = {k: inp[k].to(dev) for k in ("temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")}
    g = [t.to(dev) for t in inp["gumbel"]]
    m = toc3d_amd.build_backbone(dict(cfg, precision=precision))
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    m.alias_outputs = True
    if os.path.exists(table):
        m.load_tuning(table)
    n = toc3d_amd.build_neck(dict(configs.CPFPN_CFG, precision=precision))
    n.load_state_dict(synth.neck_state_dict(configs.CPFPN_CFG))
    n = n.to(dev).eval()
    n.alias_outputs = True
    n._tuned = m._tuned

    def step():
        return n([m(x, prev_exists=True, gumbel_noise=g, **kw).img_feats["last_feat"]])[0]
    return step


full = make(6)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    half1 = make(3)
    for _ in range(4):
        half1()
with torch.cuda.stream(s2):
    half2 = make(3)
    for _ in range(4):
        half2()
for _ in range(4):
    full()
torch.cuda.synchronize()


def t_full():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        full()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def t_halves(concurrent=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        with torch.cuda.stream(s1):
            half1()
        with torch.cuda.stream(s2 if concurrent else s1):
            half2()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


for r in range(rounds):
    a, b = t_full(), t_halves()
    print(f"round {r}: one 6-view chain {1e3 * a:.4f} ms = {1 / a:.1f} frames/s | two 3-view chains on two streams {1e3 * b:.4f} ms = {1 / b:.1f} frames/s", flush=True)
