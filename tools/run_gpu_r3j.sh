mkdir -p gpurun_out
python -m pytest tests/test_gpu_plan.py tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gumbel or bf16x3 or plan or recordings" -s 2>&1 | grep -E "bf16x3|passed|failed|Error|error|assert" | tail -16
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_bf16.py -m gpu -q --tb=short -p no:cacheprovider -k "fp32x3 or fp32x6" -s 2>&1 | grep -E "rel max|passed|failed|Error|error" | tail -30
python - <<'PY'
import json, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import toc3d_amd
from toc3d_amd import configs, synth, dist as tdist
dev = torch.device("cuda:0")
cfg = configs.get("toc3d_faster")
sd = synth.make_state_dict(cfg)
inp = synth.make_inputs(cfg, n_frames=1, views_per_frame=6, hw=(320, 800), seed=0)
d = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in inp.items()}
for prec in ("fp32x3", "fp32x6"):
    m = toc3d_amd.build_backbone(dict(cfg, precision=prec)); m.load_state_dict(sd); m = m.to(dev).eval(); m.alias_outputs = True
    n = toc3d_amd.build_neck(dict(configs.CPFPN_CFG, precision=prec)); n.load_state_dict(synth.neck_state_dict(configs.CPFPN_CFG)); n = n.to(dev).eval(); n.alias_outputs = True
    def step():
        f = m(d["x"], temp_queries=d["temp_queries"], prev_exists=True, temp_ref_points=d["temp_ref_points"], temp_vel=d["temp_vel"], temp_timestamp=d["temp_timestamp"],
              temp_ego_pose=d["temp_ego_pose"], ego_pose_inv=d["ego_pose_inv"], gumbel_noise=d["gumbel"]).img_feats["last_feat"]
        return n([f])[0]
    t0 = time.time(); step(); torch.cuda.synchronize(); t_tune = time.time() - t0
    el = tdist.timed_steps(step, 10, 3, dev)
    m._tuned.update(n._tuned)
    m.save_tuning(f"gpurun_out/toc3d_faster_320x800_{prec}.json")
    print(prec, "frames/s", 10 / el, "ms", 100 * el, "first forward (autotune) s", round(t_tune, 1), flush=True)
    del m, n
PY
