"""Development measurement: frames/s when the boundary hands over HOST images (pinned), float32 NCHW as the reference's pipeline
produces them vs uint8 HWC -- the PCIe-inclusive rate quoted in DESIGN.md, never the bench's `value`."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import toc3d_amd
from toc3d_amd import configs, synth
cfg = configs.get("toc3d_faster")
norm = dict(mean=[103.530, 116.280, 123.675], std=[57.375, 57.120, 58.395], to_rgb=False)
m = toc3d_amd.build_backbone(dict(cfg, precision="bf16", img_norm_cfg=norm)); m.load_state_dict(synth.make_state_dict(cfg)); m = m.cuda().eval()
m.view_groups, m.alias_outputs = 2, True
inp = synth.make_inputs(cfg, views_per_frame=6)
kw = {k: inp[k].cuda() for k in ("temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")}
g = [t.cuda() for t in inp["gumbel"]]
x_f = inp["x"].pin_memory()
x_u = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (6, 320, 800, 3), dtype=np.uint8)).pin_memory()
x_dev = inp["x"].cuda()
def run(src, steps=60):
    for _ in range(10): m(src() , prev_exists=True, gumbel_noise=g, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): m(src(), prev_exists=True, gumbel_noise=g, **kw)
    torch.cuda.synchronize(); return steps / (time.perf_counter() - t0)
print(f"resident float32 input      : {run(lambda: x_dev):6.1f} frames/s")
print(f"host float32 NCHW (18.4 MB) : {run(lambda: x_f.cuda(non_blocking=True)):6.1f} frames/s")
print(f"host uint8 HWC   ( 4.6 MB)  : {run(lambda: x_u.cuda(non_blocking=True)):6.1f} frames/s")
