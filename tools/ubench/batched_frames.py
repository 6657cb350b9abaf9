"""Development measurement: throughput with B frames per forward (B x 6 views as one batch, the reference's own batch dimension:
temp_queries.shape[0] = B, toc3d_eva_vit.py:230-242) against the benchmarked B = 1.  Not the headline configuration (BASELINE.json: batch 1)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import toc3d_amd
from toc3d_amd import configs, synth
cfg = configs.get("toc3d_faster")
for B in (1, 2, 3):
    m = toc3d_amd.build_backbone(dict(cfg, precision="bf16")); m.load_state_dict(synth.make_state_dict(cfg)); m = m.cuda().eval()
    m.alias_outputs = True
    t = os.path.join(os.path.dirname(toc3d_amd.__file__), "tuned", "toc3d_faster_320x800_bf16.json")
    if os.path.exists(t): m.load_tuning(t)
    neck = toc3d_amd.build_neck(dict(configs.CPFPN_CFG, precision="bf16")); neck.load_state_dict(synth.neck_state_dict(configs.CPFPN_CFG)); neck = neck.cuda().eval()
    neck.alias_outputs = True
    inp = synth.make_inputs(cfg, n_frames=B, views_per_frame=6)
    kw = {k: inp[k].cuda() for k in ("temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")}
    g = [x.cuda() for x in inp["gumbel"]]
    x = inp["x"].cuda()
    def step():
        return neck([m(x, prev_exists=True, gumbel_noise=g, **kw).img_feats["last_feat"]])[0]
    for _ in range(5): step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(30): step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 30)
    print(f"B = {B} frames per forward ({6 * B} views): {1e3 * best:.3f} ms per forward = {B / best:.1f} frames/s", flush=True)
    del m, neck
    torch.cuda.empty_cache()
