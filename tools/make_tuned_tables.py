#!/usr/bin/env python3
"""Measure and write the GEMM tile tables shipped under toc3d_amd/tuned/ (one JSON per config x resolution x precision): run on an MI355X,
    python tools/make_tuned_tables.py [outdir] [config:HxW:precision ...]
The tables map (epilogue, M, N, K) -> tile variant; every variant gives the same bits, so a table only affects speed.  bench.py loads
toc3d_amd/tuned/<config>_<H>x<W>_<precision>.json when it exists; shapes a table does not hold are tuned on first use."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import toc3d_amd
from toc3d_amd import configs, synth

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tuned"
specs = sys.argv[2:] or ["toc3d_faster:320x800:bf16:1,2", "toc3d_faster:320x800:fp32", "toc3d_fast:320x800:bf16", "eva_dense:320x800:bf16",
                         "toc3d_faster:640x1600:bf16"]
os.makedirs(out, exist_ok=True)
dev = "cuda:0"
for spec in specs:
    name, hw, prec = spec.split(":")[:3]
    batches = [int(b) for b in spec.split(":")[3].split(",")] if spec.count(":") > 2 else [1]
    H, W = (int(v) for v in hw.split("x"))
    cfg = configs.get(name)
    m = toc3d_amd.build_backbone(dict(cfg, precision=prec))
    m.load_state_dict(synth.make_state_dict(cfg))
    m = m.to(dev).eval()
    shipped = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "toc3d_amd", "tuned", f"{name}_{H}x{W}_{prec}.json")
    if os.environ.get("TOC3D_KEEP_TABLE") == "1" and os.path.exists(shipped):
        m.load_tuning(shipped)                           # keep the shipped picks, measure only the shapes the table does not hold yet
    neck = toc3d_amd.build_neck(dict(configs.CPFPN_CFG, precision=prec))
    neck.load_state_dict(synth.neck_state_dict(configs.CPFPN_CFG))
    neck = neck.to(dev).eval()
    for B in batches:                                    # frames per forward (bench.py's "batched" leg runs B = 2)
        inp = synth.make_inputs(cfg, n_frames=B, views_per_frame=6, hw=(H, W))
        x = inp["x"].to(dev)
        for _ in range(2):
            if synth.is_toc3d(cfg):
                kw = {k: inp[k].to(dev) for k in ("temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")}
                feat = m(x, prev_exists=True, gumbel_noise=[g.to(dev) for g in inp["gumbel"]], **kw).img_feats["last_feat"]
                m(x, prev_exists=False, gumbel_noise=[g.to(dev) for g in inp["gumbel"]], **kw)      # first frame of a sequence: the image-only scorer's shapes
            else:
                feat = m(x)["last_feat"]
            neck([feat])
    torch.cuda.synchronize()
    m._tuned.update(neck._tuned)
    path = os.path.join(out, f"{name}_{H}x{W}_{prec}.json")
    m.save_tuning(path)
    print(spec, len(m._tuned), "shapes ->", path, flush=True)
    del m, neck
    torch.cuda.empty_cache()
