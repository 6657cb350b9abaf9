"""Development measurement (skip-and-measure): frame time of the replayed launch plan with one thing changed at a time.
Shapes are static, so skipping a kernel (wrong values, same launches otherwise) or aliasing buffers does not change what the
remaining kernels cost -- the drop in frame time is what that kernel (or that cache miss) costs the frame *in place*.

    python tools/ubench/where_time_goes.py [tune.json]
"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import toc3d_amd
from toc3d_amd import configs, synth, lib

cfg = configs.get("toc3d_faster")
m = toc3d_amd.build_backbone(dict(cfg, precision="bf16")); m.load_state_dict(synth.make_state_dict(cfg)); m = m.cuda().eval()
m.alias_outputs = True
m.launch_mode = "plan"
tune = sys.argv[1] if len(sys.argv) > 1 else None
if tune and os.path.exists(tune):
    m.load_tuning(tune)
inp = synth.make_inputs(cfg, views_per_frame=6)
kw = {k: inp[k].cuda() for k in ("temp_queries", "temp_ref_points", "temp_vel", "temp_timestamp", "temp_ego_pose", "ego_pose_inv")}
g = [t.cuda() for t in inp["gumbel"]]
x = inp["x"].cuda()
orig_call = lib.call


def frame():
    return m(x, prev_exists=True, gumbel_noise=g, **kw)


def measure(tag, skip=(), steps=40):
    for p in (m._plans or {}).values():
        p.pop("launch", None)                       # forget recorded plans: the next frames warm up, record, replay

    def call(name, *a):
        if name in skip:
            return
        return orig_call(name, *a)
    lib.call = call
    try:
        for _ in range(4):
            frame()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                frame()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps)
    finally:
        lib.call = orig_call
    print(f"{tag:60s} {1e3 * best:7.3f} ms/frame  {1 / best:6.1f} frames/s", flush=True)
    return best


for _ in range(3):
    frame()                                         # packs, tunes
torch.cuda.synchronize()
if tune:
    m.save_tuning(tune)
base = measure("baseline (plan, 1 group)")
measure("baseline again")
for name in ("toc3d_window_attention", "toc3d_window_attention_pf", "toc3d_layernorm_act", "toc3d_layernorm_rows", "toc3d_gather_merge_ln", "toc3d_gather_merge_ln_ex", "toc3d_scatter_update",
             "toc3d_rebase_layernorm_rows", "toc3d_motion_queries", "toc3d_collapse_query_scorer", "toc3d_window_topk", "toc3d_rank_desc",
             "toc3d_score_tokens"):
    measure(f"skip {name}", skip=(name,))
measure("skip every row kernel (ln_act, ln_rows, gather, scatter, rebase)",
        skip=("toc3d_layernorm_act", "toc3d_layernorm_rows", "toc3d_gather_merge_ln", "toc3d_gather_merge_ln_ex", "toc3d_scatter_update", "toc3d_rebase_layernorm_rows"))
measure("skip all GEMMs", skip=("toc3d_linear_ex", "toc3d_linear", "toc3d_linear_fused"))
measure("skip everything but GEMMs", skip=("toc3d_window_attention", "toc3d_window_attention_pf", "toc3d_layernorm_act", "toc3d_layernorm_rows", "toc3d_gather_merge_ln", "toc3d_gather_merge_ln_ex", "toc3d_scatter_update",
                                            "toc3d_rebase_layernorm_rows", "toc3d_motion_queries", "toc3d_collapse_query_scorer", "toc3d_window_topk", "toc3d_rank_desc",
                                            "toc3d_score_tokens"))
# weights resident in the Infinity Cache: every block reads block 0's GEMM weights (25 MB instead of 600 MB per frame)
P = m._packed
saved = [dict(b) for b in P["blocks"]]
for b in P["blocks"][1:]:
    for k in ("wqkv", "wproj", "w12", "w3"):
        b[k] = P["blocks"][0][k]
measure("all blocks share block 0's GEMM weights (weights cache-resident)")
measure("  ... and only GEMMs", skip=("toc3d_window_attention", "toc3d_window_attention_pf", "toc3d_layernorm_act", "toc3d_layernorm_rows", "toc3d_gather_merge_ln", "toc3d_gather_merge_ln_ex", "toc3d_scatter_update",
                                       "toc3d_rebase_layernorm_rows", "toc3d_motion_queries", "toc3d_collapse_query_scorer", "toc3d_window_topk", "toc3d_rank_desc",
                                       "toc3d_score_tokens"))
for b, s in zip(P["blocks"], saved):
    b.update(s)
measure("baseline (weights restored)")
