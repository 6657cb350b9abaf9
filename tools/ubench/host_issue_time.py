"""Development measurement: host time to ISSUE one frame (no synchronisation inside the loop) against the wall time per frame,
per launch mode (eager: Python issues every launch; plan: one toc3d_plan_run per frame; graph: one hipGraphLaunch) and number
of view groups -- tells whether a mode is launch-bound on the host."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import toc3d_amd
from toc3d_amd import configs, synth, lib
cfg = configs.get("toc3d_faster")
m = toc3d_amd.build_backbone(dict(cfg, precision="bf16")); m.load_state_dict(synth.make_state_dict(cfg)); m = m.cuda().eval()
m.alias_outputs = True
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    m.load_tuning(sys.argv[1])
inp = synth.make_inputs(cfg, views_per_frame=6)
kw = {k: inp[k].cuda() for k in ("temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")}
g = [t.cuda() for t in inp["gumbel"]]
x = inp["x"].cuda()
for groups in (1, 2, 3):
    for mode in ("eager", "plan", "graph"):
        m.view_groups, m.launch_mode = groups, mode
        for _ in range(10): m(x, prev_exists=True, gumbel_noise=g, **kw)
        torch.cuda.synchronize()
        steps = 40
        t0 = time.perf_counter()
        for _ in range(steps): m(x, prev_exists=True, gumbel_noise=g, **kw)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # a single frame issued onto an idle GPU: issue time without back-pressure
        torch.cuda.synchronize(); u0 = time.perf_counter(); m(x, prev_exists=True, gumbel_noise=g, **kw); u1 = time.perf_counter(); torch.cuda.synchronize(); u2 = time.perf_counter()
        print(f"groups={groups} {mode:5s}: host issue {1e3 * (t1 - t0) / steps:.3f} ms/frame, wall {1e3 * (t2 - t0) / steps:.3f} ms/frame = {steps / (t2 - t0):.1f} frames/s"
              f"   | one frame on an idle GPU: issue {1e3 * (u1 - u0):.3f} ms, complete {1e3 * (u2 - u0):.3f} ms", flush=True)
if len(sys.argv) > 1:
    m.save_tuning(sys.argv[1])
