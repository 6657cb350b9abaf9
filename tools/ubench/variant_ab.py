#!/usr/bin/env python3
"""Development A/B of candidate GEMM tile variants against the shipped table, on the frame's OWN launches (real epilogue arguments):
    python tools/ubench/variant_ab.py <candidates, e.g. 70,71,72,73,74,75> [config] [HxW] [rounds]
1. one eager forward with a spy on lib.call collects the argument tuple of every distinct GEMM launch of the frame;
2. every candidate is timed on every launch as single cold launches behind a cache-sized memset (what the autotuner does), next to the table's pick;
3. the table with the winning candidates (>= 3 % faster cold) is A/B'd against the shipped table by frame time: replayed launch plans, `rounds` alternations
   in one process (cdna_hip_programming.md rule 24).
Prints per-launch cold times and the two frame-time distributions."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import toc3d_amd
from toc3d_amd import configs, lib, synth

cands = [int(v) for v in sys.argv[1].split(",")]
name = sys.argv[2] if len(sys.argv) > 2 else "toc3d_faster"
H, W = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "320x800").split("x"))
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = "cuda:0"
cfg = configs.get(name)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shipped = os.path.join(root, "toc3d_amd", "tuned", f"{name}_{H}x{W}_bf16.json")

m = toc3d_amd.build_backbone(dict(cfg, precision="bf16"))
m.load_state_dict(synth.make_state_dict(cfg))
m = m.to(dev).eval()
m.alias_outputs = True
if os.path.exists(shipped):
    m.load_tuning(shipped)
neck = toc3d_amd.build_neck(dict(configs.CPFPN_CFG, precision="bf16"))
neck.load_state_dict(synth.neck_state_dict(configs.CPFPN_CFG))
neck = neck.to(dev).eval()
neck.alias_outputs = True
neck._tuned = m._tuned
inp = synth.make_inputs(cfg, views_per_frame=6, hw=(H, W))
x = inp["x"].to(dev)
toc = synth.is_toc3d(cfg)
kw = {k: inp[k].to(dev) for k in ("temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")} if toc else {}
g = [t.to(dev) for t in inp["gumbel"]] if toc else None


def step():
    f = m(x, prev_exists=True, gumbel_noise=g, **kw).img_feats["last_feat"] if toc else m(x)["last_feat"]
    return neck([f])


def forget_plans():
    for p in (m._plans or {}).values():
        p.pop("launch", None)
    for ws in neck._ws.values():
        ws.pop("launch", None)


step()
torch.cuda.synchronize()
# ---- 1. the frame's GEMM launches ------------------------------------------------------------------------------------------------
launches = {}                # table key -> (entry point, args, index of the variant argument, current variant)
orig = lib.call


def spy(nm, *a):
    if nm == "toc3d_linear_fused":
        launches.setdefault((a[1], a[15], a[16], a[17]), (nm, a, 2, a[2]))
    elif nm == "toc3d_linear_qkv_rope":
        launches.setdefault((lib.EPI_BIAS, a[9], a[10], a[11]), (nm, a, 1, a[1]))
    return orig(nm, *a)


m.launch_mode = neck.launch_mode = "eager"
lib.call = spy
step()
lib.call = orig
m.launch_mode = neck.launch_mode = "plan"
torch.cuda.synchronize()

flush = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)


def cold(nm, a, vi, v, reps=7):
    a = list(a)
    a[vi] = v
    a[-1] = lib.stream_ptr()
    try:
        lib.call(nm, *a)
    except RuntimeError:
        return float("inf")
    ts = []
    for _ in range(reps):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.call(nm, *a)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 4]          # lower quartile: single cold launches are noisy upwards only


picks = {}
print(f"# {len(launches)} distinct GEMM launches; cold single-launch times in us (lower quartile of 7), table pick vs candidates", flush=True)
for key, (nm, a, vi, cur) in launches.items():
    t_cur = cold(nm, a, vi, cur)
    row = {v: cold(nm, a, vi, v) for v in cands}
    best = min(row, key=row.get)
    mark = ""
    if row[best] < 0.97 * t_cur:
        picks[key] = best
        mark = f"   -> v{best}"
    fl = 2.0 * key[1] * key[2] * key[3]
    print(f"epi{key[0]} M={key[1]:5d} N={key[2]:5d} K={key[3]:5d} | table v{cur:<3d} {t_cur:6.1f} ({fl / t_cur / 1e6:5.0f} TF) | "
          + " ".join(f"v{v}:{t:6.1f}" for v, t in row.items()) + mark, flush=True)

# ---- 3. frame time, shipped table vs shipped + winning candidates, alternating -----------------------------------------------------
base_tab = dict(m._tuned)
new_tab = dict(base_tab)
new_tab.update(picks)


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


if picks:
    a_t, b_t = [], []
    for _ in range(rounds):
        a_t.append(measure(base_tab))
        b_t.append(measure(new_tab))
    fmt = lambda ts: " ".join(f"{1e3 * t:.4f}" for t in ts)
    med = lambda ts: sorted(ts)[len(ts) // 2]
    print(f"# frame time ms, {rounds} alternations:  shipped table: {fmt(a_t)}  (median {1e3 * med(a_t):.4f} = {1 / med(a_t):.1f} frames/s)", flush=True)
    print(f"#                                  with {len(picks)} candidate picks: {fmt(b_t)}  (median {1e3 * med(b_t):.4f} = {1 / med(b_t):.1f} frames/s)", flush=True)
else:
    print("# no candidate beat the table's pick by 3 % on any launch: no frame-time A/B", flush=True)
