#!/usr/bin/env python3
"""Development: the frame's own gather / merge / LayerNorm launches (real selections of one ToC3D_faster frame) re-issued with every split count,
event-timed in alternating rounds.  python tools/ubench/gather_sweep.py [config] [HxW]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import toc3d_amd
from toc3d_amd import configs, lib, synth

name = sys.argv[1] if len(sys.argv) > 1 else "toc3d_faster"
H, W = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "320x800").split("x"))
dev = "cuda:0"
cfg = configs.get(name)
m = toc3d_amd.build_backbone(dict(cfg, precision="bf16", schedule=dict(launch_mode="eager")))
m.load_state_dict(synth.make_state_dict(cfg)); m = m.to(dev).eval(); m.autotune = False
inp = synth.make_inputs(cfg, views_per_frame=6, hw=(H, W))
kw = {k: inp[k].to(dev) for k in ("temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")}
calls = []
orig = lib.call
def spy(nm, *a):
    if nm == "toc3d_gather_merge_ln_split":
        calls.append(a)
    return orig(nm, *a)
lib.call = spy
m(inp["x"].to(dev), prev_exists=True, gumbel_noise=[t.to(dev) for t in inp["gumbel"]], **kw)
lib.call = orig
torch.cuda.synchronize()
flush = torch.empty(96 * 1024 * 1024, dtype=torch.float32, device=dev)
seen = {}
for a in calls:
    seen.setdefault((a[7], a[8], a[9], a[10]), a)       # (nW, N, k, rows)
print(f"# {len(calls)} gather launches per frame, {len(seen)} distinct (nW, N, k, rows); us per launch, median of 9: behind a cache-sized memset (cold) / back to back (warm)")
for key, a in seen.items():
    row = []
    for sp in ("ex", 2, 4, 8, 16):
        def run():
            if sp == "ex":
                lib.call("toc3d_gather_merge_ln_ex", *a[:18], lib.stream_ptr())
            else:
                lib.call("toc3d_gather_merge_ln_split", *a[:20], sp, lib.stream_ptr())
        run(); torch.cuda.synchronize()
        cold, warm = [], []
        for _ in range(9):
            flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); e1.synchronize()
            cold.append(e0.elapsed_time(e1) * 1e3)
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); e1.synchronize()
            warm.append(e0.elapsed_time(e1) * 1e2)
        row.append(f"{sp}: {sorted(cold)[4]:5.1f} / {sorted(warm)[4]:5.1f}")
    print(f"nW={key[0]:3d} N={key[1]:3d} k={key[2]:3d} rows={key[3]:5d} | " + "   ".join(row), flush=True)
