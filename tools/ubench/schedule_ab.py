#!/usr/bin/env python3
"""Development A/B of two launch schedules of one backbone by frame time, alternating in ONE process (cdna_hip_programming.md rule 24):
    python tools/ubench/schedule_ab.py "<schedule A as key=value,...>" "<schedule B>" [precision] [config] [HxW] [rounds] [steps] [frames per forward]
e.g.   python tools/ubench/schedule_ab.py "gather_split=True" "gather_split=False" bf16
Both models share weights, inputs and the shipped tile table; every measurement replays recorded launch plans (backbone + CPFPN neck)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import toc3d_amd
from toc3d_amd import configs, synth
from toc3d_amd import dist as tdist


def parse(txt):
    out = {}
    for kv in filter(None, txt.split(",")):
        k, v = kv.split("=")
        out[k.strip()] = {"True": True, "False": False}.get(v.strip(), int(v) if v.strip().lstrip("-").isdigit() else v.strip())
    return out


sa, sb = parse(sys.argv[1]), parse(sys.argv[2])
precision = sys.argv[3] if len(sys.argv) > 3 else "bf16"
name = sys.argv[4] if len(sys.argv) > 4 else "toc3d_faster"
H, W = (int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "320x800").split("x"))
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 5
steps = int(sys.argv[7]) if len(sys.argv) > 7 else 30
B = int(sys.argv[8]) if len(sys.argv) > 8 else 1
dev = torch.device("cuda", 0)
cfg = configs.get(name)
sd = synth.make_state_dict(cfg)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
table = os.path.join(root, "toc3d_amd", "tuned", f"{name}_{H}x{W}_{precision}.json")
inp = synth.make_inputs(cfg, n_frames=B, views_per_frame=6, hw=(H, W))
x = inp["x"].to(dev)
toc = synth.is_toc3d(cfg)
kw = {k: inp[k].to(dev) for k in ("temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")} if toc else {}
g = [t.to(dev) for t in inp["gumbel"]] if toc else None


def make(sched):
    sched = dict(sched)
    for k_ in [k_ for k_ in sched if k_.startswith("env:")]:      # "env:NAME=value": an environment variable the library reads when the plan is built (experiments only)
        os.environ[k_[4:]] = str(sched.pop(k_))
    m = toc3d_amd.build_backbone(dict(cfg, precision=precision, schedule=sched))
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
        f = m(x, prev_exists=True, gumbel_noise=g, **kw).img_feats["last_feat"] if toc else m(x)["last_feat"]
        return n([f])[0]
    for _ in range(4):
        out = step()
    torch.cuda.synchronize()
    return step, out.float().clone()


# Two instances per schedule, built in the order A, B, B, A: a model built later in the process measures 1-3 % slower than the same model built
# first (its buffers and HIP streams come later: observed with IDENTICAL schedules), so each schedule gets one early and one late instance.
first = ("A, the process's first model: shown, not part of the comparison", *make(sa))
inst = [("A", *make(sa)), ("B", *make(sb)), ("B", *make(sb)), ("A", *make(sa)), first]
out_a, out_b = inst[0][2], inst[1][2]
diff = (out_a - out_b).abs().max().item()
print(f"# A = {sa}   B = {sb}   precision {precision}, {name} {H}x{W}, {B} frame(s) per forward; max |A - B| of the neck features: {diff:.3e} "
      f"({'bit-identical' if torch.equal(out_a, out_b) else 'different bits'})", flush=True)
times = [[] for _ in inst]
for _ in range(rounds):
    for i, (_, step, _) in enumerate(inst):
        times[i].append(tdist.timed_steps(step, steps, 1, dev))
fps = lambda t: B * steps / t
med = lambda v: sorted(v)[len(v) // 2]
for i, (tag, _, _) in enumerate(inst):
    print(f"#   instance {i} ({tag}, built {'2nd 3rd 4th 5th 1st'.split()[i]}): " + " ".join(f"{fps(t):.1f}" for t in times[i]) + f"  (median {fps(med(times[i])):.1f})", flush=True)
ma = 0.5 * (med(times[0]) + med(times[3]))
mb = 0.5 * (med(times[1]) + med(times[2]))
print(f"# mean of the two instances' medians:  A {fps(ma):.1f} frames/s   B {fps(mb):.1f} frames/s   A / B = {mb / ma:.4f}", flush=True)
